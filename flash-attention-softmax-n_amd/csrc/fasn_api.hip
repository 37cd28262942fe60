// fasn_api.hip — the C ABI of libfasn (include/fasn.h): argument validation, parameter packing, dispatch.
// No allocation, no synchronisation, no environment variables: every exported symbol is declared in include/fasn.h.
#include <hip/hip_runtime.h>
#include <math.h>
#include <cxxabi.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "fasn.h"
#include "fasn_launch.h"
#include "fasn_bwd_launch.h"

using namespace fasn;

namespace fasn {
#ifdef FASN_DEV_VARIANTS
int g_bwd_variant = 0;
unsigned long long* g_timeline = nullptr;
int* g_xq = nullptr;      // developer experiment: dynamic deal of the forward's items across XCDs (fasn_fwd_kernel.h)
int g_xq_extra = 0;       // surplus workgroups per XCD of such a launch
int g_pair_mode = -1;
int g_kprot = 1;
#endif
thread_local LaunchLog* t_launch_log = nullptr;
// one line per launch: the kernel with its template arguments (demangled from the type name kernel_pretty_name<&kernel<...>> hands over:
// "fasn::KernelTag<&(void fasn::kernel<arguments>(fasn::Params))>"), grid, block, LDS
// "bf16,D=64,QB=2,plain,OCC=2,NW=4,RING=2,SEED=2" for "fasn_fwd_kernel<fasn::bf16_tag, 64, 2, 0, 2, 4, 0, 0, 2, 0, 2, 1, 0, 0>": the template
// arguments of the kernel families by NAME (flags that are off and unit factors are left out), so that a launch plan, a profile line or a
// spill table can be read without the kernel headers open. Unknown kernels: empty string.
static size_t describe_kernel(const char* name, size_t n, char* out, size_t cap) {
    struct Family { const char* base; const char* params; };   // params: names in template order; '!' prefix = omit when 0, '1' prefix = omit when 1, 'M' = mode, 'T' = element tag
    static const Family families[] = {
        {"fasn_fwd_kernel", "T D QB M OCC NW !PRIO !DROP RING !SPLIT SEED 1VH !FOLD !BF32"},
        {"fasn_bwd_dq_kernel", "T D QB M OCC !DROP DQ_SEED !BF32"},
        {"fasn_bwd_dkdv_kernel", "T D KB M OCC !DROP !GQA 1DH !BF32"},
        {"fasn_bwd_dq_ws_kernel", "T D M !DROP"},
        {"fasn_bwd_dkdv_ws_kernel", "T D M !GQA !DROP"},
        {"fasn_bwd_dq_pipe_kernel", "T M !DROP"},
        {"fasn_bwd_dkdv_pipe_kernel", "T M !DROP"},
        {"fasn_bwd_dq_ws256_kernel", "T M"},
        {"fasn_bwd_dkdv_ws256_kernel", "T M !GQA"},
        {"fasn_fwd_ws256_kernel", "T M !DROP"},
        {"fasn_bwd_delta_kernel", "T D"},
        {"fasn_fwd_combine_kernel", "T D"},
        {"fasn_bwd_dbias_ws_kernel", "T D"},
        {"fasn_bwd_dbias_kernel", "T D FAST"},
        {"fasn_f32_fwd_kernel", "D M"},
        {"fasn_f32_dq_kernel", "D M"},
        {"fasn_f32_dkdv_kernel", "D M"},
        {"fasn_f32_delta_kernel", "D"},
    };
    static const char* const modes[] = {"plain", "causal", "bias+mask", "element-load", "bias", "mask", "keypad", "bias+keypad"};
    if (cap == 0) return 0;
    out[0] = 0;
    const char* lt = (const char*)memchr(name, '<', n);
    if (lt == nullptr) return 0;
    const Family* fam = nullptr;
    for (const Family& f : families)
        if (strlen(f.base) == (size_t)(lt - name) && strncmp(f.base, name, (size_t)(lt - name)) == 0) fam = &f;
    if (fam == nullptr) return 0;
    size_t len = 0;
    auto put = [&](const char* t, size_t tn) {
        if (len + tn + 1 < cap) {
            memcpy(out + len, t, tn);
            len += tn;
            out[len] = 0;
        }
    };
    const char* a = lt + 1;
    const char* pn = fam->params;
    const char* const end = name + n;
    while (a < end && *pn) {
        const char* ae = a;   // one template argument: up to the next ',' or the closing '>' (the arguments here are flat: types and integers)
        while (ae < end && *ae != ',' && *ae != '>') ++ae;
        const char* pe = pn;
        while (*pe && *pe != ' ') ++pe;
        while (a < ae && *a == ' ') ++a;
        char val[48];
        const size_t vn = (size_t)(ae - a) < sizeof val - 1 ? (size_t)(ae - a) : sizeof val - 1;
        memcpy(val, a, vn);
        val[vn] = 0;
        const long num = strtol(val, nullptr, 10);
        char item[96];
        int in = 0;
        if (*pn == 'T' && pe - pn == 1) in = snprintf(item, sizeof item, "%s", strstr(val, "bf16") ? "bf16" : "f16");
        else if (*pn == 'M' && pe - pn == 1) in = snprintf(item, sizeof item, "%s", (num >= 0 && num < 8) ? modes[num] : val);
        else if (*pn == '!') { if (strcmp(val, "0") != 0 && strcmp(val, "false") != 0) in = snprintf(item, sizeof item, "%.*s=%s", (int)(pe - pn - 1), pn + 1, val); }
        else if (*pn == '1') { if (num != 1) in = snprintf(item, sizeof item, "%.*s=%s", (int)(pe - pn - 1), pn + 1, val); }
        else in = snprintf(item, sizeof item, "%.*s=%s", (int)(pe - pn), pn, val);
        if (in > 0) {
            if (len) put(",", 1);
            put(item, (size_t)in);
        }
        a = ae < end ? ae + 1 : end;
        pn = *pe ? pe + 1 : pe;
    }
    return len;
}

void log_launch(const char* tag_name, unsigned grid, unsigned block, int smem) {
    LaunchLog* const g = t_launch_log;
    if (g == nullptr) return;
    int st = 0;
    char* dm = abi::__cxa_demangle(tag_name, nullptr, nullptr, &st);
    const char* b = dm ? strstr(dm, "&(void ") : nullptr;
    b = b ? b + 7 : (dm ? dm : tag_name);
    if (strncmp(b, "fasn::", 6) == 0) b += 6;
    size_t n = 0;
    for (int depth = 0; b[n] != 0; ++n) {   // the name ends at the parameter list: the first '(' outside the template arguments
        if (b[n] == '<') ++depth;
        else if (b[n] == '>') --depth;
        else if (b[n] == '(' && depth == 0) break;
    }
    char cfg[192];
    describe_kernel(b, n, cfg, sizeof cfg);
    char tail[320];
    const int tn = snprintf(tail, sizeof tail, " grid=%u block=%u lds=%d cfg=%s\n", grid, block, smem, cfg[0] ? cfg : "-");
    if (g->len + n + (size_t)tn + 1 > g->cap) {   // does not fit: remember that by moving len past cap (the caller reports FASN_EINVAL)
        g->len = g->cap + 1;
    } else {
        memcpy(g->buf + g->len, b, n);
        memcpy(g->buf + g->len + n, tail, (size_t)tn);
        g->len += n + (size_t)tn;
        g->buf[g->len] = 0;
    }
    free(dm);
}
int launch_fwd_f32(const FwdParams& p, const FwdLaunch& l, hipStream_t s);
int launch_bwd_f32(const BwdParams& p, const FwdLaunch& l, hipStream_t s);
}  // namespace fasn

namespace {

bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// rows must stay 16-byte aligned: base pointer and every non-feature stride (in 2-byte elements) % 8
int check_view(const fasn_view4& v, bool required, int esize = 2) {
    if (v.ptr == nullptr) return required ? FASN_EINVAL : FASN_OK;
    if (v.stride[3] != 1) return FASN_ESTRIDE;
    if (!aligned16(v.ptr)) return FASN_EALIGN;
    for (int i = 0; i < 3; ++i)
        if (v.stride[i] % (16 / esize) != 0) return FASN_EALIGN;
    return FASN_OK;
}

// fp32 bias next to 16-bit q / k / v on the vector path (round 5): the head dims whose kernels have the fp32 image instantiation, forward and
// backward alike (head dims 32 / 64: the usual tuning points; head dim 128: the 8-wave forward with the register-staged two-buffer ring,
// fasn_fwd_d128.hip, and the ONE-wave backward kernels - the two-wave backward kernels have no LDS left for 8 KiB images; head dim 256: element
// loads). bias_vec only says that the BIAS is vector-movable: a call whose mode ends up MODE_GENERAL_SLOW for another reason (unaligned dense
// mask, fp16 scale overflow) still takes the element-load kernels, forward (launch_gen) and backward (launch_bwd_mode).
bool f32_bias_vector(int D) { return D == 32 || D == 64 || D == 128; }

int build_fwd(const fasn_fwd_args* a, FwdParams& p, FwdLaunch& l, int pass = 0) {
    if (a == nullptr) return FASN_EINVAL;
    if (a->B <= 0 || a->H <= 0 || a->Sq <= 0 || a->Sk <= 0 || a->D <= 0 || a->Dv <= 0) return FASN_EINVAL;
    if (a->dtype != FASN_DTYPE_F16 && a->dtype != FASN_DTYPE_BF16 && a->dtype != FASN_DTYPE_F32) return FASN_EDTYPE;
    if (!fasn_supported(a->dtype, a->D, a->Dv)) return FASN_EHEADDIM;
    const int esize = a->dtype == FASN_DTYPE_F32 ? 4 : 2;
    // fp32 q/k/v: the bias, if any, must be fp32 as well (FASN_BIAS_SAME means fp32 then)
    if (!(a->dropout_p >= 0.f) || a->dropout_p >= 1.f) return FASN_EINVAL;
    if (!(a->softmax_n >= 0.f) || !(a->scale >= 0.f) || !isfinite(a->scale)) return FASN_EINVAL;
    int rc;
    if ((rc = check_view(a->q, true, esize))) return rc;
    if ((rc = check_view(a->k, true, esize))) return rc;
    if ((rc = check_view(a->v, true, esize))) return rc;
    if ((rc = check_view(a->o, true, esize))) return rc;
    if (a->bias.ptr != nullptr && a->bias_dtype != FASN_BIAS_SAME && a->bias_dtype != FASN_BIAS_F32) return FASN_EINVAL;

    p.q = (const char*)a->q.ptr;
    p.k = (const char*)a->k.ptr;
    p.v = (const char*)a->v.ptr;
    p.o = (char*)a->o.ptr;
    p.lse = a->lse;
    p.mask = (const uint8_t*)a->mask.ptr;
    p.bias = (const char*)a->bias.ptr;
    for (int i = 0; i < 3; ++i) {
        p.qs[i] = a->q.stride[i];
        p.ks[i] = a->k.stride[i];
        p.vs[i] = a->v.stride[i];
        p.os[i] = a->o.stride[i];
    }
    for (int i = 0; i < 4; ++i) {
        p.ms[i] = a->mask.ptr ? a->mask.stride[i] : 0;
        p.bs[i] = a->bias.ptr ? a->bias.stride[i] : 0;
    }
    p.B = a->B;
    p.H = a->H;
    p.Sq = a->Sq;
    p.Sk = a->Sk;
    p.nqblk = 0;
    p.causal = a->causal ? 1 : 0;
    p.bias_f32 = (a->bias.ptr && (a->bias_dtype == FASN_BIAS_F32 || esize == 4)) ? 1 : 0;
    p.bias_vec = 0;
    p.mask_vec = 0;
    p.keypad_fallback = 0;
    if (a->kv_group < 0 || (a->kv_group > 1 && a->H % a->kv_group != 0)) return FASN_EINVAL;
    p.kvg = a->kv_group > 1 ? a->kv_group : 1;
    p.batch_inner = 0;
    if (a->bias.ptr) {
        const int esz = p.bias_f32 ? 4 : 2;
        const int al = 4 * esz;  // 4 keys per load
        bool ok = a->bias.stride[3] == 1 && (reinterpret_cast<uintptr_t>(a->bias.ptr) % al) == 0;
        for (int i = 0; i < 3; ++i) ok = ok && ((a->bias.stride[i] * esz) % al == 0);
        // scale 0 takes the element-load path; so does an fp32 bias next to 16-bit q / k / v unless this head dim has the fp32 image
        // instantiation (no dropout, no split-K there), and any bias next to fp32 q / k / v (their kernels have the element-load mode only)
        const bool f32_ok = !p.bias_f32 || (esize == 2 && !(a->dropout_p > 0.f) && f32_bias_vector(a->D));
        p.bias_vec = (ok && f32_ok && a->scale > 0.f) ? 1 : 0;
    }
    if (a->mask.ptr) {
        bool ok = a->mask.stride[3] == 1 && (reinterpret_cast<uintptr_t>(a->mask.ptr) % 4) == 0;
        for (int i = 0; i < 3; ++i) ok = ok && (a->mask.stride[i] % 4 == 0);
        p.mask_vec = ok ? 1 : 0;
    }
    if (a->bias.ptr && a->bias.stride[0] == 0 && a->B > 1) p.batch_inner = 1;
    p.bias_bytes = p.mask_bytes = 0;
    {   // per-(b,h) slice extents for the buffer descriptors of the vector path; slices of 2 GiB or more use the element path
        const int64_t bb = a->bias.ptr ? ((int64_t)(a->Sq - 1) * a->bias.stride[2] + a->Sk) * (p.bias_f32 ? 4 : 2) : 0;
        const int64_t mb = a->mask.ptr ? ((int64_t)(a->Sq - 1) * a->mask.stride[2] + a->Sk) : 0;
        if (bb >= (1ll << 31)) p.bias_vec = 0;
        if (mb >= (1ll << 31)) p.mask_vec = 0;
        // the range check works on whole dwords: round the extent up so an odd tail (Sk % 2 for bias, Sk % 4 for mask) is
        // still fetched; the pointer is dword aligned, so the extra bytes share a dword with valid ones and are never used
        p.bias_bytes = (unsigned)((bb + 3) & ~3ll);
        p.mask_bytes = (unsigned)((mb + 3) & ~3ll);
    }
    // dropout: 16-bit threshold, drop probability thr/65536 (the nearest representable value to dropout_p, at least 1/65536)
    p.drop_thr = 0;
    p.drop_scale = 1.f;
    if (a->dropout_p > 0.f) {
        long thr = lrintf(a->dropout_p * 65536.f);
        thr = thr < 1 ? 1 : (thr > 65535 ? 65535 : thr);
        p.drop_thr = (unsigned)thr;
        p.drop_scale = 65536.f / (65536.f - (float)thr);
    }
    p.seed_lo = (unsigned)(a->seed & 0xffffffffu) ^ ((unsigned)a->offset * 0x9E3779B1u);
    p.seed_hi = (unsigned)(a->seed >> 32) + (unsigned)(a->offset >> 32);
    p.rng = a->rng_state;   // device {seed, offset}: overrides the two by-value words when given
    if (a->rng_state != nullptr && reinterpret_cast<uintptr_t>(a->rng_state) % 8) return FASN_EALIGN;
    p.c = a->scale * kLog2e;
    {
        // extent of one (b,h) matrix = last row start + one row (a length-1 sequence may carry stride 0)
        const int64_t kb = ((int64_t)(a->Sk - 1) * a->k.stride[2] + a->D) * esize, vb = ((int64_t)(a->Sk - 1) * a->v.stride[2] + a->Dv) * esize;
        if (kb <= 0 || vb <= 0 || kb >= (1ll << 31) || vb >= (1ll << 31)) return FASN_EINVAL;  // one (b,h) K/V matrix must span < 2 GiB
        p.kbytes = (unsigned)kb;
        p.vbytes = (unsigned)vb;
    }
    p.n = a->softmax_n;

    l.dtype = a->dtype;
    l.D = a->D;
    if (a->mask.ptr || a->bias.ptr) {
        const bool vec = (!a->bias.ptr || p.bias_vec) && (!a->mask.ptr || p.mask_vec);
        l.mode = !vec ? MODE_GENERAL_SLOW : (a->bias.ptr && a->mask.ptr) ? MODE_GENERAL : a->bias.ptr ? MODE_GENERAL_B : MODE_GENERAL_M;
        // key-padding mask (one byte per key for the whole (b,h), no bias): plain kernels + a per-tile visibility word
        const bool kp_fits = (a->Sk + KT - 1) / KT <= kFwdKpMaxTiles;   // the forward kernels keep one visibility word per tile in LDS
        if (a->mask.ptr && !a->bias.ptr && a->mask.stride[2] == 0 && a->mask.stride[3] == 1 && kp_fits) {
            p.keypad_fallback = l.mode;   // what split-K and the fp32 kernels use: they have no key-padding path of their own
            l.mode = MODE_KEYPAD;
        } else if (a->mask.ptr && a->bias.ptr && p.bias_vec && a->mask.stride[2] == 0 && a->mask.stride[3] == 1 && a->dtype != FASN_DTYPE_F32 && kp_fits) {
            // vector bias + key-padding mask (ALiBi on a padded batch): bias through the vector path, mask as visibility bits
            p.keypad_fallback = l.mode;   // dropout and split-K instantiations take the dense-mask general mode instead
            l.mode = MODE_BIAS_KEYPAD;
        }
    } else {
        l.mode = a->causal ? MODE_CAUSAL : MODE_PLAIN;
    }
    // The vector kernels multiply Q (or K) by c = scale*log2e in the operand type before the MFMAs. In fp16 a large scale could
    // push an otherwise representable operand past 65504 there: such calls take the element-load kernels, which scale in fp32.
    if (a->dtype == FASN_DTYPE_F16 && fabsf(p.c) > 8.f) l.mode = MODE_GENERAL_SLOW;
    l.variant = 0;
    p.xq = nullptr;   // static deal of the items unless fasn_fwd_ws hands over counters (below)
#ifdef FASN_DEV_VARIANTS
    p.timeline = g_timeline;
    if (g_xq != nullptr && pass == 0) p.xq = g_xq;   // (developer harness, env FASN_XQ: forces the dynamic deal of the FORWARD with its own counters and surplus)
#endif
    p.pair = 0;   // set per launch (paired causal blocks, fasn_launch.h)
#ifdef FASN_DEV_VARIANTS
    p.kprot = g_kprot;
#else
    p.kprot = 1;
#endif
    p.nsplit = 1;
    p.tps = 0;
    p.part_o = nullptr;
    p.part_ml = nullptr;
    return FASN_OK;
}

// Split-K plan for short-query / long-key shapes: too few (b,h, query block) workgroups to fill 256 CUs while each one
// would walk many key tiles. Returns the number of key splits (1 = do not split) and the tiles per split (even).
int plan_splitk(const fasn_fwd_args* a, const FwdParams& p, const FwdLaunch& l, int& tps) {
    tps = 0;
    if (l.dtype == FASN_DTYPE_F32 || p.drop_thr || l.mode == MODE_GENERAL_SLOW || l.D > 128) return 1;
    if (p.bias_f32 && p.bias_vec) return 1;   // (the fp32 bias image has no split-K instantiation)
    if ((l.mode == MODE_KEYPAD || l.mode == MODE_BIAS_KEYPAD) && p.keypad_fallback == MODE_GENERAL_SLOW) return 1;
    const int64_t base_blocks = (int64_t)a->B * a->H * ((a->Sq + 127) / 128);
    int ntiles = (a->Sk + KT - 1) / KT;
    if (a->causal) {   // the last row's visible keys bound the walk
        const int kmax = a->Sk - 1;
        ntiles = kmax / KT + 1;
    }
    if (base_blocks >= 256 || ntiles < 16) return 1;
    int nsplit = (int)((1024 + base_blocks - 1) / base_blocks);
    if (nsplit > ntiles / 6) nsplit = ntiles / 6;
    if (nsplit < 2) return 1;
    tps = (ntiles + nsplit - 1) / nsplit;
    tps = (tps + 5) / 6 * 6;   // multiple of 6: the LDS buffer rotation (2 or 3 buffers) starts at buffer 0 in every split
    nsplit = (ntiles + tps - 1) / tps;
    return nsplit < 2 ? 1 : nsplit;
}
size_t splitk_bytes(const fasn_fwd_args* a, int nsplit) {
    return (size_t)a->B * a->H * nsplit * a->Sq * (size_t)(a->D + 2) * sizeof(float);
}

int dispatch_fwd(const FwdParams& p, const FwdLaunch& l, hipStream_t s) {
    if (l.dtype == FASN_DTYPE_F32) return launch_fwd_f32(p, l, s);
    switch (l.D) {
        case 32: return launch_fwd_d32(p, l, s);
        case 64: return launch_fwd_d64(p, l, s);
        case 128: return launch_fwd_d128(p, l, s);
        case 256: return launch_fwd_d256(p, l, s);
        default: return FASN_EHEADDIM;
    }
}

}  // namespace

namespace {
__global__ void rng_advance_kernel(uint64_t* state, uint64_t* out, uint64_t increment) {
    // fetch-add: two streams of one device advancing the same stream position concurrently never draw the same offset
    const uint64_t s = state[0];
    const uint64_t o = atomicAdd(reinterpret_cast<unsigned long long*>(state + 1), (unsigned long long)increment);
    if (out != nullptr) {
        out[0] = s;
        out[1] = o;
    }
}
}  // namespace

extern "C" {

int fasn_abi_version(void) { return FASN_ABI_VERSION; }

int fasn_rng_advance(uint64_t* state, uint64_t* out, uint64_t increment, fasn_stream_t stream) {
    if (state == nullptr || reinterpret_cast<uintptr_t>(state) % 8 || reinterpret_cast<uintptr_t>(out) % 8) return FASN_EINVAL;
    hipLaunchKernelGGL(rng_advance_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, state, out, increment);
    return hipGetLastError() == hipSuccess ? FASN_OK : FASN_ELAUNCH;
}

const char* fasn_strerror(int code) {
    switch (code) {
        case FASN_OK: return "ok";
        case FASN_EINVAL: return "invalid argument (null pointer, non-positive size, negative n/scale, bad enum)";
        case FASN_EDTYPE: return "unsupported element type (supported: fp16, bf16, fp32)";
        case FASN_EHEADDIM: return "unsupported head dimension (supported: D == Dv in {32, 64, 128}, and 256 for fp16 / bf16)";
        case FASN_EALIGN: return "pointer or stride breaks the 16-byte row alignment rule";
        case FASN_ESTRIDE: return "feature (last-dim) stride must be 1";
        case FASN_ELAUNCH: return "kernel launch failed";
        case FASN_EUNSUPPORTED: return "request not implemented in this build";
        case FASN_EWORKSPACE: return "workspace missing or too small";
        default: return "unknown fasn error";
    }
}

int fasn_supported(int32_t dtype, int32_t D, int32_t Dv) {
    if (dtype != FASN_DTYPE_F16 && dtype != FASN_DTYPE_BF16 && dtype != FASN_DTYPE_F32) return 0;
    if (D != Dv) return 0;
    if (D == 256) return dtype != FASN_DTYPE_F32 ? 1 : 0;   // 16-bit MFMA kernels only at this head dim
    return (D == 32 || D == 64 || D == 128) ? 1 : 0;
}

int fasn_fwd(const fasn_fwd_args* args, fasn_stream_t stream) {
    FwdParams p;
    FwdLaunch l;
    const int rc = build_fwd(args, p, l);
    if (rc) return rc;
    return dispatch_fwd(p, l, (hipStream_t)stream);
}

int fasn_fwd_path(const fasn_fwd_args* args) {
    FwdParams p;
    FwdLaunch l;
    const int rc = build_fwd(args, p, l);
    if (rc) return rc;
    if (l.dtype == FASN_DTYPE_F32) return FASN_PATH_FP32;
    int mode = l.mode;
    if (l.D > 128 && (p.drop_thr || mode == MODE_BIAS_KEYPAD)) {            // D = 256: dropout and bias + key padding through the general modes (launch_fwd_d256)
        if (mode == MODE_KEYPAD || mode == MODE_BIAS_KEYPAD) mode = p.keypad_fallback;
        if (p.drop_thr) return mode == MODE_GENERAL_SLOW ? FASN_PATH_ELEMENT : FASN_PATH_VECTOR;
    }
    switch (mode) {
        case MODE_PLAIN: case MODE_CAUSAL: return FASN_PATH_PLAIN;
        case MODE_KEYPAD: return FASN_PATH_KEYPAD;
        case MODE_BIAS_KEYPAD: return FASN_PATH_BIAS_KEYPAD;
        case MODE_GENERAL_SLOW: return FASN_PATH_ELEMENT;
        default: return FASN_PATH_VECTOR;
    }
}

// Long plain / causal launches at D = 64 deal their items dynamically across XCDs when the caller hands over 32 bytes of workspace for the eight
// counters (fasn_fwd_kernel.h: draw_item): from 8 rounds of workgroups (M0's four rounds lose 0.6 % to the memset and the atomics, 16 rounds gain 2 %).
#ifndef FASN_XQ_RULE
#define FASN_XQ_RULE 1   // (A/B: 0 = never ask for the counters: static deal everywhere, as before)
#endif
static bool xq_wanted(const FwdParams& p, const FwdLaunch& l) {
    if (!FASN_XQ_RULE) return false;
    if (l.D != 64 || l.dtype == FASN_DTYPE_F32 || p.drop_thr || ((p.B * p.H) & 7) || p.Sq < 256) return false;
    const long blocks = (long)((p.Sq + 255) / 256) * p.B * p.H;
    if (l.mode == MODE_PLAIN) return blocks >= 8 * 512;
    if (l.mode == MODE_CAUSAL) return blocks >= 2 * 8 * 512;   // (paired blocks: half as many workgroups)
    return false;
}
constexpr size_t kXqBytes = 64;

size_t fasn_fwd_workspace_bytes(const fasn_fwd_args* args) {
    FwdParams p;
    FwdLaunch l;
    if (build_fwd(args, p, l)) return 0;
    int tps;
    const int nsplit = plan_splitk(args, p, l, tps);
    if (nsplit > 1) return splitk_bytes(args, nsplit);
    return xq_wanted(p, l) ? kXqBytes : 0;
}

int fasn_fwd_ws(const fasn_fwd_args* args, void* workspace, size_t workspace_bytes, fasn_stream_t stream) {
    FwdParams p;
    FwdLaunch l;
    const int rc = build_fwd(args, p, l);
    if (rc) return rc;
    int tps;
    const int nsplit = plan_splitk(args, p, l, tps);
    if (nsplit > 1 && workspace != nullptr && workspace_bytes >= splitk_bytes(args, nsplit)) {
        if (reinterpret_cast<uintptr_t>(workspace) % 16) return FASN_EALIGN;
        p.nsplit = nsplit;
        p.tps = tps;
        p.part_o = static_cast<float*>(workspace);
        p.part_ml = p.part_o + (size_t)args->B * args->H * nsplit * args->Sq * args->D;
        return launch_fwd_splitk(p, l, (hipStream_t)stream);
    }
    if (nsplit <= 1 && workspace != nullptr && workspace_bytes >= kXqBytes && xq_wanted(p, l)) {
        if (reinterpret_cast<uintptr_t>(workspace) % 4) return FASN_EALIGN;
        p.xq = static_cast<int*>(workspace);
    }
    return dispatch_fwd(p, l, (hipStream_t)stream);
}

#ifdef FASN_DEV_VARIANTS
// developer library only (tools/libfasn_dev.so): backward A/B switch (see fasn_bwd_launch.h)
void fasn_dev_set_bwd_variant(int v) { fasn::g_bwd_variant = v; }
void fasn_dev_set_timeline(unsigned long long* buf) { fasn::g_timeline = buf; }
void fasn_dev_set_pair_mode(int v) { fasn::g_pair_mode = v; }
void fasn_dev_set_kprot(int v) { fasn::g_kprot = v; }
void fasn_dev_set_xq(int* counters, int extra) { fasn::g_xq = counters; fasn::g_xq_extra = extra; }
// developer library only (tools/libfasn_dev.so): forward with an explicit tuning variant, used by tools/fasn_harness
int fasn_fwd_variant(const fasn_fwd_args* args, fasn_stream_t stream, int variant) {
    FwdParams p;
    FwdLaunch l;
    const int rc = build_fwd(args, p, l);
    if (rc) return rc;
    l.variant = variant;
    return dispatch_fwd(p, l, (hipStream_t)stream);
}
#endif

// One-pass backward (tools/dev/fasn_bwd_fused.h): a measured loser on MI355X, kept in the DEVELOPER library only
// (tools/libfasn_dev.so, FASN_DEV_VARIANTS) for A/B work; libfasn.so ignores FASN_BWD_ONE_PASS and never asks for a workspace.
static bool bwd_fused_applies(const fasn_bwd_args* a, const FwdParams& p, const FwdLaunch& l) {
#ifndef FASN_DEV_VARIANTS
    (void)a; (void)p; (void)l;
    return false;
#else
    if (!(a->flags & FASN_BWD_ONE_PASS)) return false;
    if (l.dtype == FASN_DTYPE_F32 || l.D != 64) return false;
    if (l.mode != MODE_PLAIN && l.mode != MODE_CAUSAL) return false;
    if (p.drop_thr != 0 || p.kvg > 1) return false;
    return true;
#endif
}
static size_t bwd_fused_bytes(const fasn_bwd_args* a) { return (size_t)a->fwd.B * a->fwd.H * a->fwd.Sq * a->fwd.D * sizeof(float); }

size_t fasn_bwd_workspace_bytes(const fasn_bwd_args* args) {
    if (args == nullptr) return 0;
    FwdParams p;
    FwdLaunch l;
    if (build_fwd(&args->fwd, p, l)) return 0;
    return bwd_fused_applies(args, p, l) ? bwd_fused_bytes(args) : 0;
}

int fasn_bwd(const fasn_bwd_args* a, fasn_stream_t stream) {
    if (a == nullptr) return FASN_EINVAL;
    FwdParams fp;
    FwdLaunch l;
    int rc = build_fwd(&a->fwd, fp, l, 1);
    if (rc) return rc;
    if (a->fwd.lse == nullptr || a->delta == nullptr) return FASN_EINVAL;
    if (l.mode == MODE_KEYPAD && l.dtype == FASN_DTYPE_F32) l.mode = MODE_GENERAL_SLOW;   // (the fp32 kernels have the element-load mask path only)
    const int esize = a->fwd.dtype == FASN_DTYPE_F32 ? 4 : 2;
    if ((rc = check_view(a->dout, true, esize))) return rc;
    if ((rc = check_view(a->dq, true, esize))) return rc;
    if ((rc = check_view(a->dk, true, esize))) return rc;
    if ((rc = check_view(a->dv, true, esize))) return rc;
    BwdParams p;
    p.f = fp;
    p.dout = (const char*)a->dout.ptr;
    p.dq = (char*)a->dq.ptr;
    p.dk = (char*)a->dk.ptr;
    p.dv = (char*)a->dv.ptr;
    p.delta = a->delta;
    p.scale = a->fwd.scale;
    {
        const int64_t qb = ((int64_t)(a->fwd.Sq - 1) * a->fwd.q.stride[2] + a->fwd.D) * esize;
        const int64_t db = ((int64_t)(a->fwd.Sq - 1) * a->dout.stride[2] + a->fwd.Dv) * esize;
        if (qb <= 0 || db <= 0 || qb >= (1ll << 31) || db >= (1ll << 31)) return FASN_EINVAL;
        p.qbytes = (unsigned)qb;
        p.dobytes = (unsigned)db;
    }
    p.dqacc = nullptr;
    p.skip = 0;
    if (bwd_fused_applies(a, fp, l) && a->workspace != nullptr && a->workspace_bytes >= bwd_fused_bytes(a)) {
        if (reinterpret_cast<uintptr_t>(a->workspace) % 16) return FASN_EALIGN;
        p.dqacc = static_cast<float*>(a->workspace);
    }
    p.dbias = nullptr;
    p.dbias_vec = 0;
    for (int i = 0; i < 3; ++i) p.dbs[i] = 0;
    bool dbias_reduced = false;
    int dbias_f32 = 0;
    if (a->dbias.ptr != nullptr) {
        if (a->fwd.bias.ptr == nullptr) return FASN_EINVAL;   // nothing to differentiate
        if (a->dbias.stride[3] != 1) return FASN_ESTRIDE;
        // reduced form: the gradient summed over a broadcast batch and / or head dimension by the dbias kernel
        dbias_reduced = (a->dbias.stride[0] == 0 && a->fwd.B > 1) || (a->dbias.stride[1] == 0 && a->fwd.H > 1);
        if (dbias_reduced && (l.dtype == FASN_DTYPE_F32 || fp.drop_thr != 0 || a->dbias.stride[2] == 0)) return FASN_EUNSUPPORTED;
        dbias_f32 = (dbias_reduced && a->dbias_dtype == FASN_BIAS_F32) ? 1 : 0;
        const int osz = dbias_f32 ? 4 : esize;
        if (reinterpret_cast<uintptr_t>(a->dbias.ptr) % osz) return FASN_EALIGN;
        p.dbias = (char*)a->dbias.ptr;
        bool vec = reinterpret_cast<uintptr_t>(a->dbias.ptr) % 16 == 0;
        for (int i = 0; i < 3; ++i) {
            p.dbs[i] = a->dbias.stride[i];
            vec = vec && (a->dbias.stride[i] * osz) % 16 == 0;
        }
        p.dbias_vec = vec ? 1 : 0;
    }
    for (int i = 0; i < 3; ++i) {
        p.dos[i] = a->dout.stride[i];
        p.dqs[i] = a->dq.stride[i];
        p.dks[i] = a->dk.stride[i];
        p.dvs[i] = a->dv.stride[i];
    }
    if (l.dtype == FASN_DTYPE_F32) return launch_bwd_f32(p, l, (hipStream_t)stream);
    if (dbias_reduced) {   // dQ / dK / dV without the dense dS store, then the bias gradient by its own kernel (it needs delta)
        BwdParams pg = p;
        pg.dbias = nullptr;
        const int rc2 = launch_bwd(pg, l, (hipStream_t)stream);
        if (rc2) return rc2;
        return launch_bwd_dbias(p, l, a->dbias.stride[0] == 0 ? 1 : a->fwd.B, a->dbias.stride[1] == 0 ? 1 : a->fwd.H, dbias_f32, (hipStream_t)stream);
    }
    return launch_bwd(p, l, (hipStream_t)stream);
}

int fasn_launch_plan(const fasn_bwd_args* args, int32_t which, char* buf, size_t cap) {
    if (args == nullptr || buf == nullptr || cap == 0 || which < FASN_PLAN_FWD || which > FASN_PLAN_FWD_WS) return FASN_EINVAL;
    LaunchLog log{buf, cap, 0};
    buf[0] = 0;
    LaunchLog* const outer = t_launch_log;
    t_launch_log = &log;
    int rc;
    if (which == FASN_PLAN_FWD) rc = fasn_fwd(&args->fwd, nullptr);
    else if (which == FASN_PLAN_BWD) rc = fasn_bwd(args, nullptr);
    else rc = fasn_fwd_ws(&args->fwd, reinterpret_cast<void*>(uintptr_t(256)), ~size_t(0), nullptr);   // (nothing is launched: any aligned address stands for the workspace)
    t_launch_log = outer;
    if (rc) return rc;
    return log.len > cap ? FASN_EINVAL : (int)log.len;
}


// The family the BACKWARD of a call is routed to: asked of the launch tables themselves (a recorded plan whose dQ / dK/dV kernels are
// element-load instantiations), so it cannot drift from them. Differs from fasn_fwd_path at head dim 256, where masks other than key padding
// and every bias have vector kernels in the forward only.
int fasn_bwd_path(const fasn_bwd_args* args) {
    if (args == nullptr) return FASN_EINVAL;
    const int fp = fasn_fwd_path(&args->fwd);
    if (fp < 0 || fp == FASN_PATH_FP32 || fp == FASN_PATH_ELEMENT) return fp;
    char buf[2048];
    const int n = fasn_launch_plan(args, FASN_PLAN_BWD, buf, sizeof buf);
    if (n < 0) return n;
    return strstr(buf, "element-load") != nullptr ? FASN_PATH_ELEMENT : fp;
}

}  // extern "C"

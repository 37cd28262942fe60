// fasn_harness.cpp — standalone developer harness for libfasn (no torch): correctness against an
// in-file fp64 CPU reference, micro-probes of gfx950 instruction semantics, and hipEvent timing.
// Test infrastructure only; not part of the product path.
//   fasn_harness probe                      ds_read_b64_tr_b16 / MFMA layout probes
//   fasn_harness test                       correctness battery (fwd + bwd)
//   fasn_harness bench B H Sq Sk D dtype causal [variant] [iters] [bwd]
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <string>
#include <map>
#include <vector>
#include "fasn.h"

extern "C" int fasn_fwd_variant(const fasn_fwd_args* args, fasn_stream_t stream, int variant);
extern "C" void fasn_dev_set_xq(int* counters, int extra);   // experiment: dynamic deal of the forward's items across XCDs, `extra` surplus workgroups per XCD (env FASN_XQ)
extern "C" void fasn_dev_set_kprot(int v);       // length pairs: rotated key walk of the second element, 1 shipped / 0 off (env FASN_KPROT)
extern "C" void fasn_dev_set_pair_mode(int v);   // causal block pairing: -1 shipped rule, 0 off, 1 on (env FASN_PAIR)
extern "C" void fasn_dev_set_timeline(unsigned long long* buf);   // per-workgroup time stamps of the forward kernels (developer library)
extern "C" void fasn_dev_set_bwd_variant(int v);   // 1 = one-wave dK/dV kernel where the two-wave kernel is the default

#define HIP_CHECK(x)                                                                         \
    do {                                                                                     \
        hipError_t e_ = (x);                                                                 \
        if (e_ != hipSuccess) {                                                              \
            fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
            exit(2);                                                                         \
        }                                                                                    \
    } while (0)

// ---------------------------------------------------------------- 16-bit conversions (host)
static uint16_t f2bf(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    if ((u & 0x7fffffff) > 0x7f800000) return 0x7fc0;
    u += 0x7fff + ((u >> 16) & 1);
    return (uint16_t)(u >> 16);
}
static float bf2f(uint16_t h) {
    uint32_t u = ((uint32_t)h) << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
}
static uint16_t f2h(float f) {
    _Float16 h = (_Float16)f;
    uint16_t u;
    memcpy(&u, &h, 2);
    return u;
}
static float h2f(uint16_t u) {
    _Float16 h;
    memcpy(&h, &u, 2);
    return (float)h;
}
static uint16_t enc(float f, int dtype) { return dtype == FASN_DTYPE_BF16 ? f2bf(f) : f2h(f); }
static float dec(uint16_t u, int dtype) { return dtype == FASN_DTYPE_BF16 ? bf2f(u) : h2f(u); }

// ---------------------------------------------------------------- deterministic N(0, std^2)
struct Rng {
    uint64_t s;
    explicit Rng(uint64_t seed) : s(seed * 0x9E3779B97F4A7C15ull + 0x1234567ull) {}
    uint64_t next() {
        s ^= s << 13;
        s ^= s >> 7;
        s ^= s << 17;
        return s;
    }
    double uni() { return ((next() >> 11) + 0.5) * (1.0 / 9007199254740992.0); }
    float normal(float std) {
        double u1 = uni(), u2 = uni();
        return (float)(std * sqrt(-2.0 * log(u1)) * cos(6.283185307179586 * u2));
    }
};

struct Problem {
    int B, H, Sq, Sk, D, dtype, causal;
    float scale, n;
    int mask_kind;  // 0 none, 1 key-padding [B,1,1,Sk], 2 dense random [B,H,Sq,Sk], 3 key-padding layout with every key visible, 4 key-padding with bench.py's lengths
    int bias_kind;  // 0 none, 1 alibi [H,Sq,Sk] same dtype, 2 random f32 [B,H,Sq,Sk], 3 alibi [H,Sq,Sk] in fp32
    float std;
};

struct Host {
    std::vector<uint16_t> q, k, v, dout;
    std::vector<uint8_t> mask;
    std::vector<uint16_t> bias16;
    std::vector<float> bias32;
    int64_t ms[4] = {0, 0, 0, 0}, bs[4] = {0, 0, 0, 0};
};

static void make_inputs(const Problem& P, Host& h, uint64_t seed) {
    Rng r(seed);
    const size_t nq = (size_t)P.B * P.H * P.Sq * P.D, nk = (size_t)P.B * P.H * P.Sk * P.D;
    h.q.resize(nq);
    h.k.resize(nk);
    h.v.resize(nk);
    h.dout.resize(nq);
    for (auto& x : h.q) x = enc(r.normal(P.std), P.dtype);
    for (auto& x : h.k) x = enc(r.normal(P.std), P.dtype);
    for (auto& x : h.v) x = enc(r.normal(P.std), P.dtype);
    for (auto& x : h.dout) x = enc(r.normal(1.0f), P.dtype);
    if (P.mask_kind == 1 || P.mask_kind == 3 || P.mask_kind == 4) {
        h.mask.resize((size_t)P.B * P.Sk);
        static const int eighths[4] = {8, 7, 6, 4};   // kind 4: bench.py's keypad_mask lengths (S, 7S/8, 3S/4, S/2, repeating)
        for (int b = 0; b < P.B; ++b) {
            int valid = P.mask_kind == 3 ? P.Sk : P.mask_kind == 4 ? P.Sk * eighths[b % 4] / 8 : P.Sk - (b * P.Sk) / (2 * P.B);
            for (int j = 0; j < P.Sk; ++j) h.mask[(size_t)b * P.Sk + j] = j < valid;
        }
        h.ms[0] = P.Sk; h.ms[1] = 0; h.ms[2] = 0; h.ms[3] = 1;
    } else if (P.mask_kind == 2) {
        h.mask.resize((size_t)P.B * P.H * P.Sq * P.Sk);
        for (auto& x : h.mask) x = r.uni() < 0.7;
        h.ms[0] = (int64_t)P.H * P.Sq * P.Sk; h.ms[1] = (int64_t)P.Sq * P.Sk; h.ms[2] = P.Sk; h.ms[3] = 1;
    }
    if (P.bias_kind == 1) {
        h.bias16.resize((size_t)P.H * P.Sq * P.Sk);
        for (int hh = 0; hh < P.H; ++hh) {
            float slope = powf(2.f, -8.f * (hh + 1) / P.H);
            for (int i = 0; i < P.Sq; ++i)
                for (int j = 0; j < P.Sk; ++j)
                    h.bias16[((size_t)hh * P.Sq + i) * P.Sk + j] = enc(-slope * fabsf((float)(i + P.Sk - P.Sq - j)), P.dtype);
        }
        h.bs[0] = 0; h.bs[1] = (int64_t)P.Sq * P.Sk; h.bs[2] = P.Sk; h.bs[3] = 1;
    } else if (P.bias_kind == 3) {
        h.bias32.resize((size_t)P.H * P.Sq * P.Sk);
        for (int hh = 0; hh < P.H; ++hh) {
            float slope = powf(2.f, -8.f * (hh + 1) / P.H);
            for (int i = 0; i < P.Sq; ++i)
                for (int j = 0; j < P.Sk; ++j) h.bias32[((size_t)hh * P.Sq + i) * P.Sk + j] = -slope * fabsf((float)(i + P.Sk - P.Sq - j));
        }
        h.bs[0] = 0; h.bs[1] = (int64_t)P.Sq * P.Sk; h.bs[2] = P.Sk; h.bs[3] = 1;
    } else if (P.bias_kind == 2) {
        h.bias32.resize((size_t)P.B * P.H * P.Sq * P.Sk);
        for (auto& x : h.bias32) x = r.normal(1.0f);
        h.bs[0] = (int64_t)P.H * P.Sq * P.Sk; h.bs[1] = (int64_t)P.Sq * P.Sk; h.bs[2] = P.Sk; h.bs[3] = 1;
    }
}

// ---------------------------------------------------------------- fp64 reference for one (b,h)
struct RefOut {
    std::vector<double> o, lse, dq, dk, dv;
};
static void reference_head(const Problem& P, const Host& h, int b, int hh, bool bwd, RefOut& R) {
    const int Sq = P.Sq, Sk = P.Sk, D = P.D;
    const size_t qo = ((size_t)b * P.H + hh) * Sq * D, ko = ((size_t)b * P.H + hh) * Sk * D;
    std::vector<double> q((size_t)Sq * D), k((size_t)Sk * D), v((size_t)Sk * D), dout((size_t)Sq * D);
    for (size_t i = 0; i < q.size(); ++i) q[i] = dec(h.q[qo + i], P.dtype), dout[i] = dec(h.dout[qo + i], P.dtype);
    for (size_t i = 0; i < k.size(); ++i) k[i] = dec(h.k[ko + i], P.dtype), v[i] = dec(h.v[ko + i], P.dtype);
    R.o.assign((size_t)Sq * D, 0.0);
    R.lse.assign(Sq, 0.0);
    if (bwd) {
        R.dq.assign((size_t)Sq * D, 0.0);
        R.dk.assign((size_t)Sk * D, 0.0);
        R.dv.assign((size_t)Sk * D, 0.0);
    }
    const int coff = Sk - Sq;
    std::vector<double> x(Sk), pr(Sk);
    for (int i = 0; i < Sq; ++i) {
        double mx = -INFINITY;
        for (int j = 0; j < Sk; ++j) {
            double s = 0;
            for (int d = 0; d < D; ++d) s += q[(size_t)i * D + d] * k[(size_t)j * D + d];
            s *= P.scale;
            bool show = !(P.causal && j > i + coff);
            if (P.bias_kind == 1) s += dec(h.bias16[hh * h.bs[1] + (size_t)i * h.bs[2] + j], P.dtype);
            if (P.bias_kind >= 2) s += h.bias32[b * h.bs[0] + hh * h.bs[1] + (size_t)i * h.bs[2] + j];
            if (P.mask_kind) show = show && h.mask[b * h.ms[0] + hh * h.ms[1] + (size_t)i * h.ms[2] + j * h.ms[3]];
            x[j] = show ? s : -INFINITY;
            mx = std::max(mx, x[j]);
        }
        if (P.n > 0) mx = std::max(mx, 0.0);
        double den = 0;
        if (mx == -INFINITY) {  // fully hidden row, n == 0: library returns 0 / -inf
            R.lse[i] = -INFINITY;
            continue;
        }
        for (int j = 0; j < Sk; ++j) {
            pr[j] = exp(x[j] - mx);
            den += pr[j];
        }
        den += P.n * exp(-mx);
        R.lse[i] = mx + log(den);
        for (int j = 0; j < Sk; ++j) pr[j] /= den;
        for (int j = 0; j < Sk; ++j)
            if (pr[j] != 0)
                for (int d = 0; d < D; ++d) R.o[(size_t)i * D + d] += pr[j] * v[(size_t)j * D + d];
        if (bwd) {
            double delta = 0;
            for (int d = 0; d < D; ++d) delta += dout[(size_t)i * D + d] * R.o[(size_t)i * D + d];
            for (int j = 0; j < Sk; ++j) {
                if (pr[j] == 0) continue;
                double dp = 0;
                for (int d = 0; d < D; ++d) dp += dout[(size_t)i * D + d] * v[(size_t)j * D + d];
                const double ds = pr[j] * (dp - delta) * P.scale;
                for (int d = 0; d < D; ++d) {
                    R.dv[(size_t)j * D + d] += pr[j] * dout[(size_t)i * D + d];
                    R.dq[(size_t)i * D + d] += ds * k[(size_t)j * D + d];
                    R.dk[(size_t)j * D + d] += ds * q[(size_t)i * D + d];
                }
            }
        }
    }
}

// ---------------------------------------------------------------- device buffers
struct Dev {
    void *q = 0, *k = 0, *v = 0, *o = 0, *dout = 0, *dq = 0, *dk = 0, *dv = 0, *mask = 0, *bias = 0;
    float *lse = 0, *delta = 0;
    void* ws = 0;
    size_t ws_bytes = 0;
};
template <typename T>
static void* upload(const std::vector<T>& v) {
    void* p = nullptr;
    if (v.empty()) return p;
    HIP_CHECK(hipMalloc(&p, v.size() * sizeof(T)));
    HIP_CHECK(hipMemcpy(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
    return p;
}
static void dev_alloc(const Problem& P, const Host& h, Dev& d) {
    const size_t nq = (size_t)P.B * P.H * P.Sq * P.D, nk = (size_t)P.B * P.H * P.Sk * P.D;
    d.q = upload(h.q);
    d.k = upload(h.k);
    d.v = upload(h.v);
    d.dout = upload(h.dout);
    d.mask = upload(h.mask);
    d.bias = P.bias_kind >= 2 ? upload(h.bias32) : upload(h.bias16);
    HIP_CHECK(hipMalloc(&d.o, nq * 2));
    HIP_CHECK(hipMalloc(&d.dq, nq * 2));
    HIP_CHECK(hipMalloc(&d.dk, nk * 2));
    HIP_CHECK(hipMalloc(&d.dv, nk * 2));
    HIP_CHECK(hipMalloc((void**)&d.lse, (size_t)P.B * P.H * P.Sq * 4));
    HIP_CHECK(hipMalloc((void**)&d.delta, (size_t)P.B * P.H * P.Sq * 4));
    HIP_CHECK(hipMemset(d.o, 0xff, nq * 2));
    HIP_CHECK(hipMemset(d.dq, 0xff, nq * 2));
    HIP_CHECK(hipMemset(d.dk, 0xff, nk * 2));
    HIP_CHECK(hipMemset(d.dv, 0xff, nk * 2));
}
static void dev_free(Dev& d) {
    void* ps[] = {d.q, d.k, d.v, d.o, d.dout, d.dq, d.dk, d.dv, d.mask, d.bias, d.lse, d.delta, d.ws};
    for (void* p : ps)
        if (p) (void)hipFree(p);
}
static fasn_view4 view(void* p, int H, int S, int D) {
    fasn_view4 v;
    v.ptr = p;
    v.stride[0] = (int64_t)H * S * D;
    v.stride[1] = (int64_t)S * D;
    v.stride[2] = D;
    v.stride[3] = 1;
    return v;
}
static int g_one_pass = 0;   // harness switch: request the one-pass backward (`test ... onepass`, bench bwd_variant bit 3)
static void fill_args(const Problem& P, const Host& h, Dev& d, fasn_bwd_args& a) {
    memset(&a, 0, sizeof(a));
    fasn_fwd_args& f = a.fwd;
    f.q = view(d.q, P.H, P.Sq, P.D);
    f.k = view(d.k, P.H, P.Sk, P.D);
    f.v = view(d.v, P.H, P.Sk, P.D);
    f.o = view(d.o, P.H, P.Sq, P.D);
    f.lse = d.lse;
    if (P.mask_kind) {
        f.mask.ptr = d.mask;
        for (int i = 0; i < 4; ++i) f.mask.stride[i] = h.ms[i];
    }
    if (P.bias_kind) {
        f.bias.ptr = d.bias;
        for (int i = 0; i < 4; ++i) f.bias.stride[i] = h.bs[i];
        f.bias_dtype = P.bias_kind >= 2 ? FASN_BIAS_F32 : FASN_BIAS_SAME;
    }
    f.dtype = P.dtype;
    f.B = P.B; f.H = P.H; f.Sq = P.Sq; f.Sk = P.Sk; f.D = P.D; f.Dv = P.D;
    f.scale = P.scale;
    f.softmax_n = P.n;
    f.causal = P.causal;
    a.dout = view(d.dout, P.H, P.Sq, P.D);
    a.dq = view(d.dq, P.H, P.Sq, P.D);
    a.dk = view(d.dk, P.H, P.Sk, P.D);
    a.dv = view(d.dv, P.H, P.Sk, P.D);
    a.delta = d.delta;
    // one-pass backward (opt-in; g_one_pass): the fp32 dQ accumulator is the caller's (poisoned here: the library must clear it itself)
    a.flags = g_one_pass ? FASN_BWD_ONE_PASS : 0;
    const size_t wb = fasn_bwd_workspace_bytes(&a);
    if (wb && d.ws == nullptr) {
        HIP_CHECK(hipMalloc(&d.ws, wb));
        HIP_CHECK(hipMemset(d.ws, 0xff, wb));
        d.ws_bytes = wb;
    }
    a.workspace = d.ws;
    a.workspace_bytes = d.ws_bytes;
}

static double cmp16(const std::vector<uint16_t>& got, size_t off, const std::vector<double>& ref, int dtype, double& refmax, int& nbad) {
    double e = 0;
    for (size_t i = 0; i < ref.size(); ++i) {
        const double g = dec(got[off + i], dtype);
        if (!(g == g)) ++nbad;
        e = std::max(e, fabs(g - ref[i]));
        refmax = std::max(refmax, fabs(ref[i]));
    }
    return e;
}

static bool run_case(const char* name, Problem P, bool bwd, int variant, uint64_t seed = 1) {
    Host h;
    make_inputs(P, h, seed);
    Dev d;
    dev_alloc(P, h, d);
    fasn_bwd_args a;
    fill_args(P, h, d, a);
    int rc = fasn_fwd_variant(&a.fwd, nullptr, variant);
    if (rc) {
        printf("[FAIL] %-34s fwd rc=%d (%s)\n", name, rc, fasn_strerror(rc));
        dev_free(d);
        return false;
    }
    if (bwd) {
        rc = fasn_bwd(&a, nullptr);
        if (rc) {
            printf("[FAIL] %-34s bwd rc=%d (%s)\n", name, rc, fasn_strerror(rc));
            dev_free(d);
            return false;
        }
    }
    HIP_CHECK(hipDeviceSynchronize());
    const size_t nq = (size_t)P.B * P.H * P.Sq * P.D, nk = (size_t)P.B * P.H * P.Sk * P.D;
    std::vector<uint16_t> o(nq), dq(nq), dk(nk), dv(nk);
    std::vector<float> lse((size_t)P.B * P.H * P.Sq);
    HIP_CHECK(hipMemcpy(o.data(), d.o, nq * 2, hipMemcpyDeviceToHost));
    HIP_CHECK(hipMemcpy(lse.data(), d.lse, lse.size() * 4, hipMemcpyDeviceToHost));
    if (bwd) {
        HIP_CHECK(hipMemcpy(dq.data(), d.dq, nq * 2, hipMemcpyDeviceToHost));
        HIP_CHECK(hipMemcpy(dk.data(), d.dk, nk * 2, hipMemcpyDeviceToHost));
        HIP_CHECK(hipMemcpy(dv.data(), d.dv, nk * 2, hipMemcpyDeviceToHost));
    }
    // heads to check: all when small, else first / last
    std::vector<int> heads;
    const int nbh = P.B * P.H;
    const double cost = (double)P.Sq * P.Sk * P.D * (bwd ? 5 : 2);
    const int maxheads = std::max(1, (int)(6e9 / cost));
    if (nbh <= maxheads) for (int i = 0; i < nbh; ++i) heads.push_back(i);
    else { heads.push_back(0); if (maxheads > 1) heads.push_back(nbh - 1); if (maxheads > 2) heads.push_back(nbh / 2 + 1); }
    double eo = 0, el = 0, edq = 0, edk = 0, edv = 0, mo = 0, mdq = 0, mdk = 0, mdv = 0;
    int nbad = 0;
    std::vector<RefOut> refs(heads.size());
#pragma omp parallel for schedule(dynamic)
    for (int t = 0; t < (int)heads.size(); ++t) reference_head(P, h, heads[t] / P.H, heads[t] % P.H, bwd, refs[t]);
    for (size_t t = 0; t < heads.size(); ++t) {
        const int bh = heads[t];
        eo = std::max(eo, cmp16(o, (size_t)bh * P.Sq * P.D, refs[t].o, P.dtype, mo, nbad));
        for (int i = 0; i < P.Sq; ++i) {
            const double g = lse[(size_t)bh * P.Sq + i], r = refs[t].lse[i];
            if (r == -INFINITY) { if (g != -INFINITY) el = 1e30; }
            else el = std::max(el, fabs(g - r));
        }
        if (bwd) {
            edq = std::max(edq, cmp16(dq, (size_t)bh * P.Sq * P.D, refs[t].dq, P.dtype, mdq, nbad));
            edk = std::max(edk, cmp16(dk, (size_t)bh * P.Sk * P.D, refs[t].dk, P.dtype, mdk, nbad));
            edv = std::max(edv, cmp16(dv, (size_t)bh * P.Sk * P.D, refs[t].dv, P.dtype, mdv, nbad));
        }
    }
    // tolerance: 16-bit output rounding + P quantisation, relative to the tensor's magnitude
    const double rel = P.dtype == FASN_DTYPE_BF16 ? 2e-2 : 4e-3;
    bool ok = nbad == 0 && eo <= rel * std::max(mo, 1e-3) && el <= 2e-3;
    if (bwd) ok = ok && edq <= rel * std::max(mdq, 1e-3) && edk <= rel * std::max(mdk, 1e-3) && edv <= rel * std::max(mdv, 1e-3);
    printf("[%s] %-34s o %.2e/%.2e lse %.1e", ok ? " ok " : "FAIL", name, eo, mo, el);
    if (bwd) printf(" dq %.2e/%.2e dk %.2e/%.2e dv %.2e/%.2e", edq, mdq, edk, mdk, edv, mdv);
    printf(" nan=%d heads=%zu\n", nbad, heads.size());
    fflush(stdout);
    dev_free(d);
    return ok;
}

// ---------------------------------------------------------------- probes
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
__global__ void probe_tr(int* out) {
    __shared__ __attribute__((aligned(16))) short lds[1024];
    const int l = threadIdx.x;
    for (int i = l; i < 1024; i += 64) lds[i] = (short)i;
    __syncthreads();
    // lane l supplies the address of elements 4l..4l+3
    s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(lds + 4 * l));
    for (int j = 0; j < 4; ++j) out[l * 4 + j] = v[j];
}
__global__ void probe_mfma(float* out) {
    // A[i][k] = (7i + 3k) % 64, B[k][n] = (k == n % 16) -> C[i][n] = A[i][n%16]
    const int l = threadIdx.x;
    bf16x8 a, b;
    for (int e = 0; e < 8; ++e) {
        const int k = 8 * (l >> 5) + e;
        a[e] = (__bf16)(float)(((l & 31) * 7 + k * 3) % 64);  // small integers: exact in bf16
        b[e] = (__bf16)((k == ((l & 31) % 16)) ? 1.0f : 0.0f);
    }
    f32x16 c = {};
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
    for (int r = 0; r < 16; ++r) out[l * 16 + r] = c[r];
}
static int do_probe() {
    int* d;
    HIP_CHECK(hipMalloc((void**)&d, 256 * 4));
    hipLaunchKernelGGL(probe_tr, dim3(1), dim3(64), 0, 0, d);
    std::vector<int> h(256);
    HIP_CHECK(hipMemcpy(h.data(), d, 256 * 4, hipMemcpyDeviceToHost));
    printf("ds_read_b64_tr_b16 probe: lane l supplies elements 4l..4l+3; lane -> received element ids\n");
    int okc = 0;
    for (int l = 0; l < 64; ++l) {
        // expectation: within 16-lane group g, lane i receives element (g*64 + j*16 + i), j = 0..3
        const int g = l >> 4, i = l & 15;
        bool ok = true;
        for (int j = 0; j < 4; ++j) ok = ok && (h[l * 4 + j] == g * 64 + j * 16 + i);
        okc += ok;
        if (l < 20 || !ok) printf("  lane %2d: %4d %4d %4d %4d %s\n", l, h[l * 4], h[l * 4 + 1], h[l * 4 + 2], h[l * 4 + 3], ok ? "" : "<- UNEXPECTED");
    }
    printf("tr probe: %d/64 lanes match the expected transpose mapping\n", okc);
    float* f;
    HIP_CHECK(hipMalloc((void**)&f, 1024 * 4));
    hipLaunchKernelGGL(probe_mfma, dim3(1), dim3(64), 0, 0, f);
    std::vector<float> hf(1024);
    HIP_CHECK(hipMemcpy(hf.data(), f, 1024 * 4, hipMemcpyDeviceToHost));
    int okm = 0;
    for (int l = 0; l < 64; ++l)
        for (int r = 0; r < 16; ++r) {
            const int n = l & 31, i = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
            const float expect = (float)((i * 7 + (n % 16) * 3) % 64);
            okm += hf[l * 16 + r] == expect;
        }
    printf("mfma 32x32x16 layout probe: %d/1024 accumulator entries match (col=lane&31, row=(r&3)+8*(r>>2)+4*(lane>>5))\n", okm);
    (void)hipFree(d);
    (void)hipFree(f);
    return (okc == 64 && okm == 1024) ? 0 : 1;
}

static Problem mk(int B, int H, int Sq, int Sk, int D, int dtype, int causal, float n, int mk_ = 0, int bk = 0, float scale = -1.f) {
    Problem P;
    P.B = B; P.H = H; P.Sq = Sq; P.Sk = Sk; P.D = D; P.dtype = dtype; P.causal = causal;
    P.scale = scale < 0 ? 1.0f / sqrtf((float)D) : scale;
    P.n = n; P.mask_kind = mk_; P.bias_kind = bk; P.std = 0.5f;
    return P;
}

static int do_test(int variant, bool quick) {
    int fails = 0;
    const int BF = FASN_DTYPE_BF16, HF = FASN_DTYPE_F16;
    struct C { const char* name; Problem P; bool bwd; };
    std::vector<C> cases = {
        {"d64 bf16 256x256 n1", mk(2, 2, 256, 256, 64, BF, 0, 1.f), true},
        {"d64 f16 256x256 n0", mk(1, 2, 256, 256, 64, HF, 0, 0.f), true},
        {"d64 bf16 512x512 causal n1", mk(1, 2, 512, 512, 64, BF, 1, 1.f), true},
        {"d64 f16 384x640 causal n.5", mk(1, 2, 384, 640, 64, HF, 1, 0.5f), true},
        {"d64 bf16 ragged 100x77 n4", mk(2, 1, 100, 77, 64, BF, 0, 4.f), true},
        {"d64 f16 ragged 3x5 causal n1", mk(2, 1, 3, 5, 64, HF, 1, 1.f), true},
        {"d64 bf16 causal Sq>Sk 300x200", mk(1, 2, 300, 200, 64, BF, 1, 0.f), true},
        {"d128 bf16 256x320 n.5", mk(1, 2, 256, 320, 128, BF, 0, 0.5f), true},
        {"d128 f16 257x129 causal n1", mk(1, 2, 257, 129, 128, HF, 1, 1.f), true},
        {"d32 f16 512x512 n1", mk(2, 2, 512, 512, 32, HF, 0, 1.f), true},
        {"d32 bf16 200x333 causal n0", mk(1, 2, 200, 333, 32, BF, 1, 0.f), true},
        {"d64 bf16 keypad mask n1", mk(2, 2, 256, 256, 64, BF, 0, 1.f, 1, 0), true},
        {"d64 f16 dense mask causal n.5", mk(1, 2, 200, 264, 64, HF, 1, 0.5f, 2, 0), true},
        {"d64 bf16 alibi n.5", mk(2, 4, 256, 256, 64, BF, 0, 0.5f, 0, 1), true},
        {"d128 bf16 alibi+keypad n.5", mk(2, 4, 320, 320, 128, BF, 0, 0.5f, 1, 1), true},
        {"d128 bf16 alibi+keypad lengths 8/7/6/4 (4,8,1024) n.5 (length pairs, rotated second walk)", mk(4, 8, 1024, 1024, 128, BF, 0, 0.5f, 4, 1), true},
        {"d128 f16 alibi+keypad lengths (3,8,576x832) n1 (odd batch)", mk(3, 8, 576, 832, 128, HF, 0, 1.f, 4, 1), true},
        {"d64 f16 f32bias+mask causal n1", mk(1, 2, 130, 190, 64, HF, 1, 1.f, 2, 2), true},
        {"d64 bf16 f32 bias (vector image) n1", mk(2, 4, 256, 320, 64, BF, 0, 1.f, 0, 2), true},
        {"d64 f16 f32 bias + dense mask causal n.5 (vector image)", mk(1, 2, 200, 264, 64, HF, 1, 0.5f, 2, 2), true},
        {"d64 bf16 f32 bias + key padding n1 (vector image + visibility bits)", mk(3, 2, 320, 448, 64, BF, 0, 1.f, 1, 2), true},
        {"d32 bf16 f32 bias n1 (vector image)", mk(2, 2, 256, 256, 32, BF, 0, 1.f, 0, 2), true},
        {"d64 bf16 f32 alibi [H,L,S] + key padding lengths n.5", mk(4, 8, 512, 512, 64, BF, 0, 0.5f, 4, 3), true},
        {"d128 bf16 f32 alibi [H,L,S] + key padding lengths n.5 (4-wave forward, one-wave backward)", mk(4, 8, 384, 512, 128, BF, 0, 0.5f, 4, 3), true},
        {"d128 f16 f32 bias + dense mask causal n1", mk(1, 2, 200, 264, 128, HF, 1, 1.f, 2, 2), true},
        {"d128 bf16 f32 bias n0", mk(2, 2, 256, 320, 128, BF, 0, 0.f, 0, 2), true},
        {"d64 bf16 scale.3 n4", mk(1, 1, 1024, 1152, 64, BF, 0, 4.f, 0, 0, 0.3f), true},
        {"d256 bf16 256x320 n.5", mk(1, 2, 256, 320, 256, BF, 0, 0.5f), true},
        {"d256 f16 257x129 causal n1", mk(1, 2, 257, 129, 256, HF, 1, 1.f), true},
        {"d256 bf16 keypad n1", mk(2, 2, 200, 264, 256, BF, 0, 1.f, 1, 0), true},
        {"d256 bf16 alibi+keypad n.5", mk(2, 4, 320, 320, 256, BF, 0, 0.5f, 1, 1), true},
        {"d256 f16 f32bias+mask causal n1", mk(1, 2, 130, 190, 256, HF, 1, 1.f, 2, 2), true},
        {"d64 bf16 1100x1300 causal n1", mk(1, 2, 1100, 1300, 64, BF, 1, 1.f), true},
        {"d64 f16 1300x1100 causal n0", mk(1, 2, 1300, 1100, 64, HF, 1, 0.f), true},
        {"d64 f16 33x1000 n1", mk(2, 2, 33, 1000, 64, HF, 0, 1.f), true},
        {"d64 bf16 (4,16,2048) causal n1 (paired blocks)", mk(4, 16, 2048, 2048, 64, BF, 1, 1.f), true},
        {"d64 f16 (2,4,1000x1536) causal n0", mk(2, 4, 1000, 1536, 64, HF, 1, 0.f), true},
        {"d64 bf16 1x64 n1", mk(1, 1, 1, 64, 64, BF, 0, 1.f), true},
        {"d64 bf16 65x1 causal n1", mk(1, 2, 65, 1, 64, BF, 1, 1.f), true},
    };
    if (!quick) {
        cases.push_back({"d64 bf16 (8,16,1024) n1", mk(8, 16, 1024, 1024, 64, BF, 0, 1.f), true});
        cases.push_back({"d64 f16 (8,16,4096) causal n1", mk(8, 16, 4096, 4096, 64, HF, 1, 1.f), false});
        cases.push_back({"d64 bf16 (8,16,4096) n1", mk(8, 16, 4096, 4096, 64, BF, 0, 1.f), false});
        cases.push_back({"d128 bf16 (1,4,2048) n.5", mk(1, 4, 2048, 2048, 128, BF, 0, 0.5f), true});
    }
    for (auto& c : cases) fails += !run_case(c.name, c.P, c.bwd, variant);
    // spike test: one query/key pair with a huge score late in the sequence forces a rescale
    {
        Problem P = mk(1, 1, 256, 512, 64, BF, 0, 1.f);
        P.std = 0.5f;
        // handled through seed variation only (random data); explicit spike below
        fails += !run_case("d64 bf16 seed7", P, true, variant, 7);
    }
    printf("%s: %d failing case(s)\n", fails ? "TESTS FAILED" : "ALL TESTS PASSED", fails);
    return fails ? 1 : 0;
}

static int do_bench(int argc, char** argv) {
    if (argc < 9) {
        fprintf(stderr, "bench B H Sq Sk D dtype(0=f16,1=bf16) causal [variant] [iters] [bwd] [n] [mask_kind] [bias_kind] [bwd_variant]\n");
        return 2;
    }
    Problem P = mk(atoi(argv[2]), atoi(argv[3]), atoi(argv[4]), atoi(argv[5]), atoi(argv[6]), atoi(argv[7]), atoi(argv[8]), 1.f);
    const int variant = argc > 9 ? atoi(argv[9]) : 0;
    const int iters = argc > 10 ? atoi(argv[10]) : 20;
    const bool bwd = argc > 11 ? atoi(argv[11]) != 0 : false;
    if (argc > 12) P.n = (float)atof(argv[12]);
    if (argc > 13) P.mask_kind = atoi(argv[13]);
    if (argc > 14) P.bias_kind = atoi(argv[14]);
    if (argc > 15) {   // bit 3 (8): request the one-pass backward
        fasn_dev_set_bwd_variant(atoi(argv[15]));
        g_one_pass = (atoi(argv[15]) >> 3) & 1;
    }
    Host h;
    make_inputs(P, h, 3);
    Dev d;
    dev_alloc(P, h, d);
    fasn_bwd_args a;
    fill_args(P, h, d, a);
    hipEvent_t e0, e1;
    HIP_CHECK(hipEventCreate(&e0));
    HIP_CHECK(hipEventCreate(&e1));
    for (int i = 0; i < 5; ++i) {
        int rc = fasn_fwd_variant(&a.fwd, nullptr, variant);
        if (rc) { printf("fwd rc=%d\n", rc); return 1; }
    }
    HIP_CHECK(hipDeviceSynchronize());
    HIP_CHECK(hipEventRecord(e0, 0));
    for (int i = 0; i < iters; ++i) fasn_fwd_variant(&a.fwd, nullptr, variant);
    HIP_CHECK(hipEventRecord(e1, 0));
    HIP_CHECK(hipEventSynchronize(e1));
    float ms = 0;
    HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
    ms /= iters;
    double flops = 4.0 * P.B * P.H * (double)P.Sq * P.Sk * P.D;
    if (P.causal) flops *= 0.5;
    printf("fwd B%d H%d Sq%d Sk%d D%d %s causal%d variant%d: %.4f ms  %.1f TFLOP/s (%.1f%% of 2500)\n", P.B, P.H, P.Sq, P.Sk, P.D,
           P.dtype ? "bf16" : "f16", P.causal, variant, ms, flops / ms * 1e-9, flops / ms * 1e-9 / 25.0);
    if (bwd) {
        for (int i = 0; i < 3; ++i) fasn_bwd(&a, nullptr);
        HIP_CHECK(hipDeviceSynchronize());
        HIP_CHECK(hipEventRecord(e0, 0));
        for (int i = 0; i < iters; ++i) fasn_bwd(&a, nullptr);
        HIP_CHECK(hipEventRecord(e1, 0));
        HIP_CHECK(hipEventSynchronize(e1));
        HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
        ms /= iters;
        printf("bwd: %.4f ms  %.1f TFLOP/s (2.5x fwd flops)\n", ms, 2.5 * flops / ms * 1e-9);
    }
    fflush(stdout);
    dev_free(d);
    return 0;
}

// timeline B H Sq Sk D dtype causal [variant]: one forward launch with per-workgroup time stamps (100 MHz clock): where a workgroup's
// life goes (prologue / tile loop / epilogue), how well the CU slots stay covered, how long the ramp and the tail are
static int do_timeline(int argc, char** argv) {
    if (argc < 9) return 2;
    Problem P = mk(atoi(argv[2]), atoi(argv[3]), atoi(argv[4]), atoi(argv[5]), atoi(argv[6]), atoi(argv[7]), atoi(argv[8]), 1.f);
    const int variant = argc > 9 ? atoi(argv[9]) : 0;
    if (argc > 10) P.n = (float)atof(argv[10]);
    if (argc > 11) P.mask_kind = atoi(argv[11]);
    if (argc > 12) P.bias_kind = atoi(argv[12]);
    Host h;
    make_inputs(P, h, 3);
    Dev d;
    dev_alloc(P, h, d);
    fasn_bwd_args a;
    fill_args(P, h, d, a);
    const size_t maxwg = (size_t)P.B * P.H * ((P.Sq + 31) / 32) * 2;
    unsigned long long* tl = nullptr;
    HIP_CHECK(hipMalloc(&tl, maxwg * 64));
    for (int i = 0; i < 5; ++i) fasn_fwd_variant(&a.fwd, nullptr, variant);
    HIP_CHECK(hipDeviceSynchronize());
    HIP_CHECK(hipMemset(tl, 0, maxwg * 64));
    fasn_dev_set_timeline(tl);
    fasn_fwd_variant(&a.fwd, nullptr, variant);
    fasn_fwd_variant(&a.fwd, nullptr, variant);   // the second of two back-to-back launches is the one kept
    HIP_CHECK(hipDeviceSynchronize());
    fasn_dev_set_timeline(nullptr);
    std::vector<unsigned long long> t(maxwg * 8);
    HIP_CHECK(hipMemcpy(t.data(), tl, maxwg * 64, hipMemcpyDeviceToHost));
    size_t n = 0;
    while (n < maxwg && t[n * 8 + 3] != 0) ++n;
    if (const char* dump = getenv("FASN_TIMELINE_DUMP")) {   // raw stamps for offline analysis (tools/timeline_gaps.py)
        FILE* f = fopen(dump, "wb");
        if (f) { fwrite(t.data(), 64, n, f); fclose(f); }
    }
    if (n == 0) { printf("no stamps\n"); return 1; }
    unsigned long long tmin = ~0ull, tmax = 0;
    for (size_t i = 0; i < n; ++i) { tmin = std::min(tmin, t[i * 8]); tmax = std::max(tmax, t[i * 8 + 3]); }
    const double tick = 0.01;   // us
    std::vector<double> pro(n), epi(n), per(n), life(n);
    double sum_life = 0, sum_pro = 0, sum_epi = 0, sum_loop = 0, sum_tiles = 0;
    std::map<unsigned, int> cus;
    for (size_t i = 0; i < n; ++i) {
        const unsigned long long* w = &t[i * 8];
        pro[i] = (w[1] - w[0]) * tick; epi[i] = (w[3] - w[2]) * tick; life[i] = (w[3] - w[0]) * tick;
        per[i] = w[6] ? (w[2] - w[1]) * tick / (double)w[6] : 0;
        sum_life += life[i]; sum_pro += pro[i]; sum_epi += epi[i]; sum_loop += (w[2] - w[1]) * tick; sum_tiles += (double)w[6];
        const unsigned hw = (unsigned)w[4], xcc = (unsigned)w[5] & 15;
        cus[(xcc << 16) | (hw & 0xff00) | ((hw >> 13) & 7) << 4 | ((hw >> 12) & 1)]++;   // (xcc, cu, se, sh)
    }
    auto pct = [&](std::vector<double> v, double q) { std::sort(v.begin(), v.end()); return v[(size_t)(q * (v.size() - 1))]; };
    const double span = (tmax - tmin) * tick;
    printf("timeline B%d H%d Sq%d Sk%d D%d causal%d variant%d: %zu workgroups on %zu CUs, span %.1f us\n", P.B, P.H, P.Sq, P.Sk, P.D, P.causal, variant, n, cus.size(), span);
    printf("  per workgroup: prologue mean %.2f us (p10 %.2f p50 %.2f p90 %.2f)   epilogue mean %.2f (p50 %.2f p90 %.2f)   tile mean %.3f us (p10 %.3f p50 %.3f p90 %.3f)\n",
           sum_pro / n, pct(pro, .1), pct(pro, .5), pct(pro, .9), sum_epi / n, pct(epi, .5), pct(epi, .9), sum_loop / sum_tiles, pct(per, .1), pct(per, .5), pct(per, .9));
    printf("  sum of workgroup lifetimes %.0f us = %.2f resident workgroups per CU over the span; prologue %.1f%% loop %.1f%% epilogue %.1f%% of the lifetimes\n",
           sum_life, sum_life / span / cus.size(), 100 * sum_pro / sum_life, 100 * sum_loop / sum_life, 100 * sum_epi / sum_life);
    // concurrency profile: resident workgroups in 20 slices of the span
    printf("  resident workgroups / CU per 5%% slice of the span:");
    for (int sl = 0; sl < 20; ++sl) {
        const double a0 = tmin * tick + span * sl / 20, a1 = a0 + span / 20;
        double acc = 0;
        for (size_t i = 0; i < n; ++i) {
            const double s0 = std::max(a0, t[i * 8] * tick), s1 = std::min(a1, t[i * 8 + 3] * tick);
            if (s1 > s0) acc += s1 - s0;
        }
        printf(" %.2f", acc / (span / 20) / cus.size());
    }
    printf("\n");
    // start-to-start: first start of each workgroup relative to the kernel's first stamp
    std::vector<double> st(n);
    for (size_t i = 0; i < n; ++i) st[i] = (t[i * 8] - tmin) * tick;
    std::sort(st.begin(), st.end());
    const size_t first = std::min(n, cus.size() * 3);
    printf("  workgroup starts: #%zu at %.1f us, #%zu at %.1f us; last start %.1f us; last end %.1f us\n", cus.size(), st[std::min(n, cus.size()) - 1], first, st[first - 1], st[n - 1], span);
    fflush(stdout);
    hipFree(tl);
    dev_free(d);
    return 0;
}

int main(int argc, char** argv) {
    if (argc < 2) {
        fprintf(stderr, "usage: %s probe | test [variant] [quick] | bench ...\n", argv[0]);
        return 2;
    }
    if (const char* pm = getenv("FASN_PAIR")) fasn_dev_set_pair_mode(atoi(pm));
    if (const char* kr = getenv("FASN_KPROT")) fasn_dev_set_kprot(atoi(kr));
    if (const char* xq = getenv("FASN_XQ")) {
        int* ctr = nullptr;
        if (hipMalloc(&ctr, 64) == hipSuccess) fasn_dev_set_xq(ctr, atoi(xq));
    }
    if (const char* bv = getenv("FASN_BWDV")) fasn_dev_set_bwd_variant(atoi(bv));   // backward kernel variant for `test` (bench takes it as an argument)
    std::string cmd = argv[1];
    if (cmd == "probe") return do_probe();
    if (cmd == "test") {
        g_one_pass = argc > 4 && atoi(argv[4]) != 0;   // test [variant] [quick] [one_pass]
        return do_test(argc > 2 ? atoi(argv[2]) : 0, argc > 3 && atoi(argv[3]) != 0);
    }
    if (cmd == "bench") return do_bench(argc, argv);
    if (cmd == "timeline") return do_timeline(argc, argv);
    return 2;
}

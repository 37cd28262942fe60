// Gradient of a batch- / head-broadcast additive bias, reduced in the kernel (fasn_bwd_dbias.h): launch plumbing.
#include "fasn_bwd_launch.h"
#include "fasn_bwd_dbias.h"
#include "fasn_bwd_dbias_ws.h"
namespace fasn {

template <typename Tag, int D, bool FAST>
static int launch_k(const DbiasParams& dp, hipStream_t s) {
    constexpr int smem = dbias_smem_bytes(D);
    constexpr auto kern = &fasn_bwd_dbias_kernel<Tag, D, FAST>;
    ensure_smem<kern>(smem);
    const long tiles = (long)dp.Bb * dp.Hb * dp.nqb * dp.nkb;
    const long grid = tiles < 1024 ? tiles : 1024;   // persistent: one workgroup per CU is resident (LDS), four waves of tiles each
    FASN_LAUNCH(kern, dim3((unsigned)grid), dim3(256), smem, s, dp);
    return launch_rc();
}
// the two-role pipeline (fasn_bwd_dbias_ws.h): one persistent workgroup of 8 waves per CU, its own lean parameter block
template <typename Tag, int D>
static int launch_ws(const DbiasParams& dp, hipStream_t s) {
    constexpr int smem = dbias_ws_smem_bytes<D>();
    constexpr auto kern = &fasn_bwd_dbias_ws_kernel<Tag, D>;
    ensure_smem<kern>(smem);
    // CUs of the CURRENT device (the stream's device by the ABI contract), cached per device ordinal like ensure_smem's attribute bit:
    // a partitioned part (CPX: 32 CUs per device) next to a whole one must not inherit its grid
    static std::atomic<int> cus[64];
    int dev = 0, n = 0;
    if (t_launch_log != nullptr || hipGetDevice(&dev) != hipSuccess) dev = -1;   // (recording a launch plan: no HIP call, a whole MI355X assumed)
    if (dev >= 0 && dev < 64) n = cus[dev].load(std::memory_order_relaxed);
    if (n == 0) {
        if (dev < 0 || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        if (dev >= 0 && dev < 64) cus[dev].store(n, std::memory_order_relaxed);
    }
    const BwdParams& b = dp.b;
    const FwdParams& f = b.f;
    DbwParams w;
    w.q = f.q, w.k = f.k, w.v = f.v, w.dout = b.dout, w.bias = f.bias, w.mask = f.mask, w.lse = f.lse, w.delta = b.delta, w.dbias = b.dbias;
    w.qs0 = f.qs[0], w.qs1 = f.qs[1], w.ks0 = f.ks[0], w.ks1 = f.ks[1], w.vs0 = f.vs[0], w.vs1 = f.vs[1];
    w.dos0 = b.dos[0], w.dos1 = b.dos[1], w.bs0 = f.bs[0], w.bs1 = f.bs[1], w.ms0 = f.ms[0], w.ms1 = f.ms[1], w.dbs0 = b.dbs[0], w.dbs1 = b.dbs[1];
    w.qs2 = (int)f.qs[2], w.ks2 = (int)f.ks[2], w.vs2 = (int)f.vs[2], w.dos2 = (int)b.dos[2], w.bs2 = (int)f.bs[2], w.dbs2 = (int)b.dbs[2];
    w.qbytes = b.qbytes, w.dobytes = b.dobytes, w.kbytes = f.kbytes, w.vbytes = f.vbytes, w.bias_bytes = f.bias_bytes, w.mask_bytes = f.mask_bytes;
    w.B = f.B, w.H = f.H, w.Sq = f.Sq, w.Sk = f.Sk, w.causal = f.causal, w.c = f.c;
    w.Bb = dp.Bb, w.Hb = dp.Hb, w.nqb = dp.nqb, w.nkb = dp.nkb;
    const long tiles = (long)dp.Bb * dp.Hb * dp.nqb * dp.nkb;
    const int grid = (int)(tiles < n ? tiles : n);
    w.dk = grid % dp.nkb, w.dq = (grid / dp.nkb) % dp.nqb, w.dh = (grid / (dp.nkb * dp.nqb)) % dp.Hb, w.db = grid / (dp.nkb * dp.nqb * dp.Hb);
    // XCD-local tile order: written for 8 XCDs of 32 CUs with workgroup i on XCD i % 8 (a whole MI355X); any other device keeps the plain order
    w.xorder = (n == 256 && grid == 256 && dp.Hb % 8 == 0 && dp.nkb % 8 == 0 && dp.nqb % 4 == 0 && !(FASN_BWD_VARIANT & 4096)) ? 1 : 0;   // (developer library: bit 12 = plain tile order)
    FASN_LAUNCH(kern, dim3((unsigned)grid), dim3(512), smem, s, w);
    return launch_rc();
}
template <typename Tag, int D>
static int launch_one(const DbiasParams& dp, hipStream_t s) {
    const FwdParams& f = dp.b.f;
    if constexpr (D == 64 || D == 128) {
        // 16-bit bias and gradient whose rows move in 16-byte pieces; no mask, or one without a row dimension (key padding) and unit key stride
        const bool ws = f.bias_vec && !f.bias_f32 && f.bs[3] == 1 && !dp.out_f32 && dp.b.dbias_vec && (f.mask == nullptr || (f.ms[2] == 0 && f.ms[3] == 1)) && f.kvg == 1 &&
                        !(FASN_BWD_VARIANT & 2048);   // (developer library: bwd_variant bit 11 = the round-3 kernel, for A/B)
        if (ws) return launch_ws<Tag, D>(dp, s);
    }
    // the instantiation without per-element global access: 16-bit bias and gradient whose rows move in 16-byte pieces, a mask (if
    // any) whose rows move in dwords (for a key-padding mask: the row stride is 0)
    const bool fast = f.bias_vec && !f.bias_f32 && f.bs[3] == 1 && !dp.out_f32 && dp.b.dbias_vec && (f.mask == nullptr || (f.mask_vec && f.ms[3] == 1));
    return fast ? launch_k<Tag, D, true>(dp, s) : launch_k<Tag, D, false>(dp, s);
}
template <typename Tag>
static int launch_d(const DbiasParams& dp, int D, hipStream_t s) {
    switch (D) {
        case 32: return launch_one<Tag, 32>(dp, s);
        case 64: return launch_one<Tag, 64>(dp, s);
        case 128: return launch_one<Tag, 128>(dp, s);
        case 256: return launch_one<Tag, 256>(dp, s);
        default: return -3;
    }
}
// p.dbias / p.dbs describe the [Bb,Hb,Sq,Sk] output; needs p.delta (the caller launches this after the backward's delta kernel)
int launch_bwd_dbias(const BwdParams& p, const FwdLaunch& l, int Bb, int Hb, int out_f32, hipStream_t s) {
    DbiasParams dp;
    dp.b = p;
    dp.Bb = Bb;
    dp.Hb = Hb;
    dp.out_f32 = out_f32;
    dp.nqb = (p.f.Sq + 127) / 128;
    dp.nkb = (p.f.Sk + 127) / 128;
    return l.dtype == 1 ? launch_d<bf16_tag>(dp, l.D, s) : launch_d<f16_tag>(dp, l.D, s);
}

}  // namespace fasn

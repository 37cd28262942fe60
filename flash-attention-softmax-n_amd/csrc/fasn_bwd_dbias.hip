// Gradient of a batch- / head-broadcast additive bias, reduced in the kernel (fasn_bwd_dbias.h): launch plumbing.
#include "fasn_bwd_launch.h"
#include "fasn_bwd_dbias.h"
namespace fasn {

template <typename Tag, int D, bool FAST>
static int launch_k(const DbiasParams& dp, hipStream_t s) {
    constexpr int smem = dbias_smem_bytes(D);
    constexpr auto kern = &fasn_bwd_dbias_kernel<Tag, D, FAST>;
    ensure_smem<kern>(smem);
    const long tiles = (long)dp.Bb * dp.Hb * dp.nqb * dp.nkb;
    const long grid = tiles < 1024 ? tiles : 1024;   // persistent: one workgroup per CU is resident (LDS), four waves of tiles each
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(256), smem, s, dp);
    return launch_rc();
}
template <typename Tag, int D>
static int launch_one(const DbiasParams& dp, hipStream_t s) {
    const FwdParams& f = dp.b.f;
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

// One-pass backward (fasn_bwd_fused.h): launch plumbing. A translation unit of its own: the kernel is the one under active tuning.
#include "fasn_bwd_launch.h"
#include "fasn_bwd_fused.h"
namespace fasn {

// delta, zero the fp32 dQ accumulator, the fused kernel, round dQ. D = 64, plain / causal.
template <typename Tag, int ABL = 0>
static int launch_bwd_fused(BwdParams p, int mode, hipStream_t s) {
    constexpr int D = 64;
    const int nbh = p.f.B * p.f.H;
    const int64_t rows = (int64_t)nbh * p.f.Sq;
    constexpr int RPB = 256 / (D / 8);
    FASN_LAUNCH((fasn_bwd_delta_kernel<Tag, D>), dim3((unsigned)((rows + RPB - 1) / RPB)), dim3(256), 0, s, p);
    if (t_launch_log == nullptr && hipMemsetAsync(p.dqacc, 0, (size_t)rows * D * sizeof(float), s) != hipSuccess) return -6;   // (fasn_launch_plan records, it touches no device memory)
    constexpr int smem = fused_smem_bytes();
    p.nblk = (p.f.Sk + FBN - 1) / FBN;
    if (mode == MODE_CAUSAL) {
        constexpr auto kern = &fasn_bwd_fused_kernel<Tag, MODE_CAUSAL, ABL>;
        ensure_smem<kern>(smem);
        FASN_LAUNCH(kern, dim3((unsigned)(p.nblk * nbh)), dim3(512), smem, s, p);
    } else {
        constexpr auto kern = &fasn_bwd_fused_kernel<Tag, MODE_PLAIN, ABL>;
        ensure_smem<kern>(smem);
        FASN_LAUNCH(kern, dim3((unsigned)(p.nblk * nbh)), dim3(512), smem, s, p);
    }
    FASN_LAUNCH((fasn_bwd_dq_convert_kernel<Tag, D>), dim3((unsigned)((rows + RPB - 1) / RPB)), dim3(256), 0, s, p);
    return launch_rc();
}

int launch_bwd_fused_d64(const BwdParams& p, const FwdLaunch& l, hipStream_t s) {
#ifdef FASN_DEV_VARIANTS
    // developer ablations (results are NOT gradients): bwd_variant >> 4 = 1 no atomics, 2 no dQ GEMM, 3 no dQ GEMM and no dS image
    switch ((FASN_BWD_VARIANT >> 4) & 3) {
        case 1: return launch_bwd_fused<bf16_tag, 1>(p, l.mode, s);
        case 2: return launch_bwd_fused<bf16_tag, 2>(p, l.mode, s);
        case 3: return launch_bwd_fused<bf16_tag, 3>(p, l.mode, s);
        default: break;
    }
#endif
    return l.dtype == 1 ? launch_bwd_fused<bf16_tag>(p, l.mode, s) : launch_bwd_fused<f16_tag>(p, l.mode, s);
}


#ifdef FASN_DEV_VARIANTS
// developer experiment: the split backward with bigger key blocks per wave in the dK/dV kernel (plain mode, bf16)
int launch_bwd_d64_exp(const BwdParams& p, int which, hipStream_t s) {
    switch (which) {
        case 1: return launch_bwd_one<bf16_tag, 64, 1, 2, MODE_PLAIN, 2, 2>(p, s);   // 64 keys per wave, two waves per SIMD
        case 2: return launch_bwd_one<bf16_tag, 64, 1, 2, MODE_PLAIN, 2, 1>(p, s);   // 64 keys per wave, register budget of one wave per SIMD
        default: return launch_bwd_one<bf16_tag, 64, 1, 4, MODE_PLAIN, 2, 1>(p, s);  // 128 keys per wave, one wave per SIMD
    }
}
#endif
}  // namespace fasn

// Split-K forward ("decode" shapes: few query rows, many keys): key ranges of one (b,h, query block) on separate workgroups
// + the combine kernel. Instantiations: QB = 1, plain / causal / vector general (an absent mask or bias is a zero-range
// descriptor / an all-ones word), every head dimension.
#include "fasn_launch.h"
namespace fasn {
namespace {
template <typename Tag, int D, int MODE>
int launch_splitk_one(FwdParams p, hipStream_t s) {
    constexpr int BM = 128;
    // memory-bound: D <= 64: K/V go straight to LDS, two tiles ahead (three buffers of 8-16 KiB, three workgroups per CU);
    // D = 128: register staging, two LDS buffers, so that two workgroups (2 x 64 KiB) fit on a CU
    constexpr int RING = (D == 128 && MODE != MODE_GENERAL) ? 0 : 2;   // (the D = 128 mask/bias kernel spills with staging registers)
    constexpr int OCC = 2;
    constexpr int smem = fwd_smem(D, RING, MODE, 4, 1);
    p.nqblk = (p.Sq + BM - 1) / BM;
    constexpr auto kern = &fasn_fwd_kernel<Tag, D, 1, MODE, OCC, 4, 0, 0, RING, 1>;
    ensure_smem<kern>(smem);
    FASN_LAUNCH(kern, dim3((unsigned)(p.nqblk * p.nsplit * p.B * p.H)), dim3(256), smem, s, p);
    const int64_t nthr = (int64_t)p.B * p.H * p.Sq * (D / 4);
    FASN_LAUNCH((fasn_fwd_combine_kernel<Tag, D>), dim3((unsigned)((nthr + 255) / 256)), dim3(256), 0, s, p);
    return launch_rc();
}
template <typename Tag, int D>
int launch_splitk_mode(const FwdParams& p, int mode, hipStream_t s) {
    switch (mode) {
        case MODE_PLAIN: return launch_splitk_one<Tag, D, MODE_PLAIN>(p, s);
        case MODE_CAUSAL: return launch_splitk_one<Tag, D, MODE_CAUSAL>(p, s);
        case MODE_GENERAL:
        case MODE_GENERAL_B:
        case MODE_GENERAL_M:
        case MODE_KEYPAD:
        case MODE_BIAS_KEYPAD: return launch_splitk_one<Tag, D, MODE_GENERAL>(p, s);   // (the plan requires the vector-mask alignment)
        default: return -7;
    }
}
template <typename Tag>
int launch_splitk_d(const FwdParams& p, const FwdLaunch& l, hipStream_t s) {
    switch (l.D) {
        case 32: return launch_splitk_mode<Tag, 32>(p, l.mode, s);
        case 64: return launch_splitk_mode<Tag, 64>(p, l.mode, s);
        case 128: return launch_splitk_mode<Tag, 128>(p, l.mode, s);
        default: return -3;
    }
}
}  // namespace
int launch_fwd_splitk(const FwdParams& p, const FwdLaunch& l, hipStream_t s) {
    return l.dtype == 1 ? launch_splitk_d<bf16_tag>(p, l, s) : launch_splitk_d<f16_tag>(p, l, s);
}
}  // namespace fasn

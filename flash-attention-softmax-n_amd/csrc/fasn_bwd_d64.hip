// D = 64 backward instantiations: <QB (dQ: 32-row blocks/wave), KB (dK/dV: 32-key blocks/wave), occupancies>
// (the two-wave kernels of D = 128 measured slower here: (8,16,4096,64) backward 2.18 ms against 1.81 ms - at D = 64 the one-wave
// kernels already run two waves per SIMD and the exponentials, not registers, are the limit)
#include "fasn_bwd_launch.h"
namespace fasn {
#ifdef FASN_DEV_VARIANTS
int launch_bwd_d64_exp(const BwdParams& p, int which, hipStream_t s);
#endif
int launch_bwd_d64(const BwdParams& p, const FwdLaunch& l, hipStream_t s) {
#ifdef FASN_DEV_VARIANTS
    if (((FASN_BWD_VARIANT >> 8) & 3) && l.mode == MODE_PLAIN && l.dtype == 1) return launch_bwd_d64_exp(p, (FASN_BWD_VARIANT >> 8) & 3, s);
#endif
    if (p.dqacc != nullptr)   // fasn_api.hip sets the accumulator only where the one-pass backward applies (plain / causal, no dropout, no GQA)
        return launch_bwd_fused_d64(p, l, s);
    return l.dtype == 1 ? launch_bwd_mode<bf16_tag, 64, 1, 1, 2, 2>(p, l.mode, s) : launch_bwd_mode<f16_tag, 64, 1, 1, 2, 2>(p, l.mode, s);
}
}  // namespace fasn

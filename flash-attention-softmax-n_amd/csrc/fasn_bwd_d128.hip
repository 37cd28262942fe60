// D = 128 backward instantiations: <QB (dQ: 32-row blocks/wave), KB (dK/dV: 32-key blocks/wave), occupancies, WS>
// WS = 1: dK / dV by the two-wave kernel (fasn_bwd_dkdv_ws.h); dropout and the element-load mode keep the one-wave kernel
#include "fasn_bwd_launch.h"
namespace fasn {
int launch_bwd_d128(const BwdParams& p, const FwdLaunch& l, hipStream_t s) {
    return l.dtype == 1 ? launch_bwd_mode<bf16_tag, 128, 1, 1, 2, 1, 1>(p, l.mode, s) : launch_bwd_mode<f16_tag, 128, 1, 1, 2, 1, 1>(p, l.mode, s);
}
}  // namespace fasn

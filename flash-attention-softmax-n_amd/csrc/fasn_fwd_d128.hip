// D = 128 forward instantiations (QB=1: 128 query rows per workgroup; O^T alone is 64 registers).
#include "fasn_launch.h"
namespace fasn {
template <typename Tag>
static int launch_gen(const FwdParams& p, const FwdLaunch& l, hipStream_t s) {
    switch (l.mode) {
        case MODE_GENERAL: return launch_fwd_one<Tag, 128, 1, MODE_GENERAL, 2, 8>(p, s);
        case MODE_GENERAL_B: return launch_fwd_one<Tag, 128, 1, MODE_GENERAL_B, 2, 8>(p, s);
        case MODE_GENERAL_M: return launch_fwd_one<Tag, 128, 1, MODE_GENERAL_M, 2, 8>(p, s);
        default: return launch_fwd_one<Tag, 128, 1, MODE_GENERAL_SLOW, 1>(p, s);
    }
}
template <typename Tag>
static int go(const FwdParams& p, const FwdLaunch& l, hipStream_t s) {
    if (p.drop_thr) return launch_fwd_drop<Tag, 128, 1, 1>(p, l.mode, s);
    if (l.mode >= MODE_GENERAL) return launch_gen<Tag>(p, l, s);
    if (l.variant == 1) return launch_fwd_mode<Tag, 128, 1, 1>(p, l.mode, s);
    return launch_fwd_mode<Tag, 128, 1, 2>(p, l.mode, s);
}
int launch_fwd_d128(const FwdParams& p, const FwdLaunch& l, hipStream_t s) {
    return l.dtype == 1 ? go<bf16_tag>(p, l, s) : go<f16_tag>(p, l, s);
}
}  // namespace fasn

// D = 64 forward instantiations. variant selects the tuning point (internal; ABI callers get variant 0):
//   0: QB=2 (64 rows/wave, 256 rows/WG), 2 waves/SIMD     1: QB=1 (128 rows/WG), 3 waves/SIMD
//   2: QB=2, 1 wave/SIMD (512 registers)                  3: QB=1, 2 waves/SIMD
#include "fasn_launch.h"
namespace fasn {
template <typename Tag>
static int go(const FwdParams& p, const FwdLaunch& l, hipStream_t s) {
    const bool gen = l.mode == MODE_GENERAL;
    switch (l.variant) {
        case 1: return gen ? launch_fwd_one<Tag, 64, 1, MODE_GENERAL, 1>(p, s) : launch_fwd_mode<Tag, 64, 1, 3>(p, l.mode, s);
        case 2: return gen ? launch_fwd_one<Tag, 64, 2, MODE_GENERAL, 1>(p, s) : launch_fwd_mode<Tag, 64, 2, 1>(p, l.mode, s);
        case 3: return gen ? launch_fwd_one<Tag, 64, 1, MODE_GENERAL, 1>(p, s) : launch_fwd_mode<Tag, 64, 1, 2>(p, l.mode, s);
        default: return gen ? launch_fwd_one<Tag, 64, 2, MODE_GENERAL, 1>(p, s) : launch_fwd_mode<Tag, 64, 2, 2>(p, l.mode, s);
    }
}
int launch_fwd_d64(const FwdParams& p, const FwdLaunch& l, hipStream_t s) {
    return l.dtype == 1 ? go<bf16_tag>(p, l, s) : go<f16_tag>(p, l, s);
}
}  // namespace fasn

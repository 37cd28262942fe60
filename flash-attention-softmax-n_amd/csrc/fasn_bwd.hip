#include "fasn_bwd_launch.h"
namespace fasn {
int launch_bwd(const BwdParams& p, const FwdLaunch& l, hipStream_t s) {
    switch (l.D) {
        case 32: return launch_bwd_d32(p, l, s);
        case 64: return launch_bwd_d64(p, l, s);
        case 128: return launch_bwd_d128(p, l, s);
        case 256: return launch_bwd_d256(p, l, s);
        default: return -3;
    }
}
}  // namespace fasn

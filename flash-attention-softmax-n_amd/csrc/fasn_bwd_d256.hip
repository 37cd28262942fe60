// D = 256 backward instantiations: the one-wave dQ and dK/dV kernels with the whole register file (one wave per SIMD:
// dQ^T is 128 accumulator registers next to 128 of Q / dO fragments; dK^T + dV^T would be 256 next to 128 of K / V fragments, so
// the dK/dV kernel runs as two workgroups per key block that each own half of the features, DH = 2).
// Key padding rides the plain kernels; dense masks, bias and dropout take the element-load kernels.
#include "fasn_bwd_launch.h"
namespace fasn {
template <typename Tag>
static int go(const BwdParams& p, int mode, hipStream_t s) {
    if (mode == MODE_BIAS_KEYPAD) mode = p.f.keypad_fallback;   // bias + key padding: the dense-mask view of the same mask
    if (p.f.drop_thr) return launch_bwd_one<Tag, 256, 1, 1, MODE_GENERAL_SLOW, 1, 1, 1, 0, 2>(p, s);
    switch (mode) {
        case MODE_CAUSAL: return launch_bwd_one<Tag, 256, 1, 1, MODE_CAUSAL, 1, 1, 0, 0, 2>(p, s);
        case MODE_PLAIN:   // the key-padding instantiation without a mask (every key visible): the plain one spills at this head dim
        case MODE_KEYPAD: return launch_bwd_one<Tag, 256, 1, 1, MODE_KEYPAD, 1, 1, 0, 0, 2>(p, s);
        // dense masks / bias: the element-load kernels (the dK/dV kernel's additive tile next to four 32 KiB Q / dO buffers would
        // need 161 KiB of LDS)
        default: return launch_bwd_one<Tag, 256, 1, 1, MODE_GENERAL_SLOW, 1, 1, 0, 0, 2>(p, s);
    }
}
int launch_bwd_d256(const BwdParams& p, const FwdLaunch& l, hipStream_t s) {
    return l.dtype == 1 ? go<bf16_tag>(p, l.mode, s) : go<f16_tag>(p, l.mode, s);
}
}  // namespace fasn

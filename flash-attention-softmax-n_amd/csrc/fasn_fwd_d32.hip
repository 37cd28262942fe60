// D = 32 forward instantiations (QB=2: 256 query rows per workgroup).
#include "fasn_launch.h"
namespace fasn {
template <typename Tag>
static int launch_gen(const FwdParams& p, const FwdLaunch& l, hipStream_t s) {
    if (p.bias_f32) {   // fp32 bias next to 16-bit q / k / v: the fp32 image instantiations
        switch (l.mode) {
            case MODE_GENERAL: return launch_fwd_one<Tag, 32, 2, MODE_GENERAL, 1, 4, 0, 2, 0, 1, 0, 1>(p, s);
            case MODE_GENERAL_B: return launch_fwd_one<Tag, 32, 2, MODE_GENERAL_B, 1, 4, 0, 2, 0, 1, 0, 1>(p, s);
            case MODE_BIAS_KEYPAD: return launch_fwd_one<Tag, 32, 2, MODE_BIAS_KEYPAD, 1, 4, 0, 2, 0, 1, 0, 1>(p, s);
            default: break;
        }
    }
    // (round 6: compiled for TWO waves per SIMD, direct-to-LDS - 178 - 207 registers, no spill; round 5 compiled them for one wave per SIMD,
    // 324 registers; profiles/r06_d32_vector_modes_two_waves_per_simd_ab.log)
#ifndef FASN_D32_VEC_OCC
#define FASN_D32_VEC_OCC 2
#endif
#ifndef FASN_D32_VEC_RING
#define FASN_D32_VEC_RING 2
#endif
    // causal next to a mask / bias (unequal workgroups) on a grid of less than FASN_D32_VEC_CAUSAL_BLOCKS 256-row blocks: 128-row workgroups, three per CU
    // (111 - 140 registers) - a one-round launch of 256-row blocks takes as long as the call without the causal flag
#ifndef FASN_D32_VEC_QB1_ALL
#define FASN_D32_VEC_QB1_ALL 0
#endif
#ifndef FASN_D32_VEC_CAUSAL_BLOCKS
#define FASN_D32_VEC_CAUSAL_BLOCKS (1L << 40)   // (every causal launch: -20 .. -28 % up to one round of 256-row blocks, still -3 % at 2048 blocks; profiles/r06_causal_next_to_a_bias_forward_rule_ab.log)
#endif
    // The same 128-row workgroups without the causal flag (same log): bias + dense mask -6 .. -8 % at every size; bias + key padding (length pairs: half
    // the workgroups) -26 / -15 % below 1024 blocks, +3.5 % above; bias alone and mask alone: a tie - they keep 256 rows.
    const long b256 = (long)((p.Sq + 255) / 256) * p.B * p.H;
    const bool small_wg = FASN_D32_VEC_QB1_ALL || (p.causal && b256 < FASN_D32_VEC_CAUSAL_BLOCKS) || l.mode == MODE_GENERAL || (l.mode == MODE_BIAS_KEYPAD && b256 < 1024);
    if (small_wg) {
        switch (l.mode) {
            case MODE_GENERAL: return launch_fwd_one<Tag, 32, 1, MODE_GENERAL, 3, 4, 2, 2>(p, s);
            case MODE_GENERAL_B: return launch_fwd_one<Tag, 32, 1, MODE_GENERAL_B, 3, 4, 2, 2>(p, s);
            case MODE_GENERAL_M: return launch_fwd_one<Tag, 32, 1, MODE_GENERAL_M, 3, 4, 2, 2>(p, s);
            case MODE_BIAS_KEYPAD: return launch_fwd_one<Tag, 32, 1, MODE_BIAS_KEYPAD, 3, 4, 2, 2>(p, s);
            default: break;
        }
    }
    switch (l.mode) {
        case MODE_GENERAL: return launch_fwd_one<Tag, 32, 2, MODE_GENERAL, FASN_D32_VEC_OCC, 4, FASN_D32_VEC_RING, 2>(p, s);
        case MODE_GENERAL_B: return launch_fwd_one<Tag, 32, 2, MODE_GENERAL_B, FASN_D32_VEC_OCC, 4, FASN_D32_VEC_RING, 2>(p, s);
        case MODE_GENERAL_M: return launch_fwd_one<Tag, 32, 2, MODE_GENERAL_M, FASN_D32_VEC_OCC, 4, FASN_D32_VEC_RING, 2>(p, s);
        case MODE_BIAS_KEYPAD: return launch_fwd_one<Tag, 32, 2, MODE_BIAS_KEYPAD, FASN_D32_VEC_OCC, 4, FASN_D32_VEC_RING, 2>(p, s);
        default: return launch_fwd_one<Tag, 32, 2, MODE_GENERAL_SLOW, 1>(p, s);
    }
}
template <typename Tag>
static int go(const FwdParams& p, const FwdLaunch& l, hipStream_t s) {
    if (p.drop_thr) return launch_fwd_drop<Tag, 32, 2, 2>(p, l.mode, s);
    if (l.mode >= MODE_GENERAL && l.mode != MODE_KEYPAD) return launch_gen<Tag>(p, l, s);
#ifdef FASN_DEV_VARIANTS
    if (l.variant == 80) return launch_fwd_cfg<Tag, 32, 2, 2, 4, 0, 2>(p, l.mode, s);   // seeded accumulators + packed row sums
    if (l.variant == 81) return launch_fwd_cfg<Tag, 32, 2, 2, 4, 2, 2>(p, l.mode, s);
    if (l.variant == 1) return launch_fwd_mode<Tag, 32, 2, 2>(p, l.mode, s);   // unseeded, for A/B (740 vs 796 TFLOP/s at (8,16,4096,32))
    if (l.variant == 82) return launch_fwd_cfg<Tag, 32, 2, 2, 4, 1, 2>(p, l.mode, s);   // two-set ring (800 vs 833 TFLOP/s for the unrolled direct-to-LDS loop)
#endif
    return launch_fwd_cfg<Tag, 32, 2, 2, 4, 2, 2>(p, l.mode, s);
}
int launch_fwd_d32(const FwdParams& p, const FwdLaunch& l, hipStream_t s) {
    return l.dtype == 1 ? go<bf16_tag>(p, l, s) : go<f16_tag>(p, l, s);
}
}  // namespace fasn

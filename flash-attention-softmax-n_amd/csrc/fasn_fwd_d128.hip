// D = 128 forward instantiations (QB=1: 32 query rows per wave; O^T alone is 64 registers).
// A/B tuning points and ablations: FASN_DEV_VARIANTS builds only (tools/libfasn_dev.so), see fasn_launch.h.
#include "fasn_launch.h"
#ifndef FASN_BF32_4WAVE
#define FASN_BF32_4WAVE 0   // (round 5 A/B: the first fp32-bias forward at D = 128 - 4 waves, one workgroup per CU - instead of the 8-wave register-staged one)
#endif
namespace fasn {
template <typename Tag>
static int launch_gen(const FwdParams& p, const FwdLaunch& l, hipStream_t s) {
#ifdef FASN_DEV_VARIANTS
    if (l.variant == 1) {   // A/B: unseeded
        switch (l.mode) {
            case MODE_GENERAL: case MODE_GENERAL_B: return launch_fwd_one<Tag, 128, 1, MODE_GENERAL, 2, 8, 2>(p, s);
            case MODE_GENERAL_M: return launch_fwd_one<Tag, 128, 1, MODE_GENERAL_M, 2, 8, 2>(p, s);
            default: break;
        }
    }
#endif
    if (p.bias_f32) {
        // fp32 bias next to 16-bit q / k / v: the fp32 image instantiations. Eight 8 KiB images do not fit next to the three K/V buffers of the
        // direct-to-LDS ring (160 KiB + the visibility words), but they do next to the TWO buffers of the register-staged ring (64 + 64 KiB):
        // the 8-wave workgroup with RING = 0 (round 5; 242 - 256 registers, no spill). FASN_BF32_4WAVE=1: the first build (4 waves, one
        // workgroup per CU, direct-to-LDS), for A/B.
#if FASN_BF32_4WAVE
        switch (l.mode) {
            case MODE_GENERAL: return launch_fwd_one<Tag, 128, 1, MODE_GENERAL, 1, 4, 2, 2, 0, 1, 0, 1>(p, s);
            case MODE_GENERAL_B: return launch_fwd_one<Tag, 128, 1, MODE_GENERAL_B, 1, 4, 2, 2, 0, 1, 0, 1>(p, s);
            case MODE_BIAS_KEYPAD: return launch_fwd_one<Tag, 128, 1, MODE_BIAS_KEYPAD, 1, 4, 2, 2, 0, 1, 0, 1>(p, s);
            default: break;
        }
#else
        switch (l.mode) {
            case MODE_GENERAL: return launch_fwd_one<Tag, 128, 1, MODE_GENERAL, 2, 8, 0, 2, 0, 1, 0, 1>(p, s);
            case MODE_GENERAL_B: return launch_fwd_one<Tag, 128, 1, MODE_GENERAL_B, 2, 8, 0, 2, 0, 1, 0, 1>(p, s);
            case MODE_BIAS_KEYPAD: return launch_fwd_one<Tag, 128, 1, MODE_BIAS_KEYPAD, 2, 8, 0, 2, 0, 1, 0, 1>(p, s);
            default: break;
        }
#endif
    }
    switch (l.mode) {
        case MODE_GENERAL: return launch_fwd_one<Tag, 128, 1, MODE_GENERAL, 2, 8, 2, 2>(p, s);
        case MODE_GENERAL_B:   // the bias-only instantiation spills (37 VGPRs) at D = 128: the bias + key-padding kernel with every key kept
                               // (no mask image to handle: 5.32 vs 5.68 ms at (1,128,8192,128)), or the bias + mask kernel for very long key ranges
            if ((p.Sk + KT - 1) / KT <= kFwdKpMaxTiles) return launch_fwd_one<Tag, 128, 1, MODE_BIAS_KEYPAD, 2, 8, 2, 2>(p, s);
            return launch_fwd_one<Tag, 128, 1, MODE_GENERAL, 2, 8, 2, 2>(p, s);
        case MODE_GENERAL_M: return launch_fwd_one<Tag, 128, 1, MODE_GENERAL_M, 2, 8, 2, 2>(p, s);
        case MODE_BIAS_KEYPAD: return launch_fwd_one<Tag, 128, 1, MODE_BIAS_KEYPAD, 2, 8, 2, 2>(p, s);
        default: return launch_fwd_one<Tag, 128, 1, MODE_GENERAL_SLOW, 1>(p, s);
    }
}
template <typename Tag>
static int go(const FwdParams& p, const FwdLaunch& l, hipStream_t s) {
    if (p.drop_thr) return launch_fwd_drop<Tag, 128, 1, 1>(p, l.mode, s);
    if (l.mode >= MODE_GENERAL && l.mode != MODE_KEYPAD) return launch_gen<Tag>(p, l, s);   // key-padding masks ride the plain tuning points
#ifdef FASN_DEV_VARIANTS
    if (l.variant == 1) return launch_fwd_mode<Tag, 128, 1, 1>(p, l.mode, s);
    // A/B tuning points (tools/fasn_harness bench ... <variant>)
    if (l.variant == 40) return launch_fwd_ring<Tag, 128, 1, 2>(p, l.mode, s);
    if (l.variant == 43) return launch_fwd_ring<Tag, 128, 1, 2, 2>(p, l.mode, s);
    if (l.variant == 13) return launch_fwd_cfg<Tag, 128, 1, 2, 8, 0>(p, l.mode, s);   // 8 waves share one K/V tile
    if (l.variant == 14) return launch_fwd_cfg<Tag, 128, 1, 2, 8, 2>(p, l.mode, s);   // + direct-to-LDS staging
    if (l.variant == 15) return launch_fwd_cfg<Tag, 128, 1, 2, 8, 1>(p, l.mode, s);
    if (l.variant == 16) return launch_fwd_cfg<Tag, 128, 2, 1, 4, 1>(p, l.mode, s);   // 64 rows per wave, one wave per SIMD
    if (l.variant == 17) return launch_fwd_cfg<Tag, 128, 2, 1, 4, 0>(p, l.mode, s);
    if (l.variant == 18) return launch_fwd_cfg<Tag, 128, 2, 1, 4, 2>(p, l.mode, s);
    if (l.variant == 80) return launch_fwd_cfg<Tag, 128, 1, 2, 8, 1, 2>(p, l.mode, s);   // seeded accumulators + packed row sums
    if (l.variant == 81) return launch_fwd_cfg<Tag, 128, 1, 2, 8, 2, 2>(p, l.mode, s);
    if (l.variant == 82) return launch_fwd_cfg<Tag, 128, 1, 2, 4, 0, 2>(p, l.mode, s);   // 4-wave kernels (small grids)
    if (l.variant == 83) return launch_fwd_cfg<Tag, 128, 1, 2, 4, 2, 2>(p, l.mode, s);
    if (l.variant == 84) return launch_fwd_cfg<Tag, 128, 1, 2, 4, 0, 0>(p, l.mode, s);   // = the small-grid default, for A/B
#endif
    // auto: with enough work to give every CU two 256-row blocks, one 8-wave workgroup per CU (eight waves share each staged
    // K/V tile, tiles loaded two ahead in two register sets) beats two 4-wave workgroups: 1134 vs 1014 TFLOP/s at
    // (4,32,8192,128) bf16, 886 vs 790 at (2,16,2048,128); staging + barrier cost 26 % of the 4-wave kernel at D = 128
    const long blocks256 = (long)((p.Sq + 255) / 256) * p.B * p.H;
    // Both with seeded accumulators (Q pre-scaled, S starts at -m) and packed row sums: 1137 vs 1093 TFLOP/s at C4's shape,
    // 680 vs 631 at (1,16,2048,128) where the 4-wave kernel with direct-to-LDS staging runs.
    // The 8-wave kernel stages K/V straight to LDS with the loop unrolled by its three buffers: 1164 vs 1112 TFLOP/s for the two-set
    // register ring (variant 80), causal 1034 vs 984.
    if (blocks256 >= 512 && p.Sq >= 256) return launch_fwd_cfg<Tag, 128, 1, 2, 8, 2, 2>(p, l.mode, s);
    return launch_fwd_cfg<Tag, 128, 1, 2, 4, 2, 2>(p, l.mode, s);
}
int launch_fwd_d128(const FwdParams& p, const FwdLaunch& l, hipStream_t s) {
    return l.dtype == 1 ? go<bf16_tag>(p, l, s) : go<f16_tag>(p, l, s);
}
}  // namespace fasn

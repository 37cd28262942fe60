// D = 64 forward instantiations. libfasn.so compiles the production tuning points only; the other tuning points of the same
// kernel (staging scheme, rows per wave, waves per workgroup) exist in FASN_DEV_VARIANTS builds (tools/libfasn_dev.so) for A/B
// runs through tools/fasn_harness and fasn_fwd_variant - nothing in the shipped library can reach them.
#include "fasn_launch.h"
namespace fasn {
template <typename Tag>
static int launch_gen(const FwdParams& p, const FwdLaunch& l, hipStream_t s) {
#ifdef FASN_DEV_VARIANTS
    if (l.variant == 70) return launch_fwd_one<Tag, 64, 2, MODE_GENERAL, 2>(p, s);          // A/B: 64 rows per wave
    if (l.variant == 71) return launch_fwd_one<Tag, 64, 1, MODE_GENERAL, 2, 8>(p, s);       // A/B: 8 waves
    if (l.variant == 72) return launch_fwd_one<Tag, 64, 1, MODE_GENERAL, 2, 4, 2>(p, s);    // A/B: direct K/V staging
    if (l.variant == 73) return launch_fwd_one<Tag, 64, 2, MODE_GENERAL, 2, 4, 2>(p, s);
    if (l.variant == 1) {   // A/B: unseeded
        switch (l.mode) {
            case MODE_GENERAL: return launch_fwd_one<Tag, 64, 1, MODE_GENERAL, 2>(p, s);
            case MODE_GENERAL_B: return launch_fwd_one<Tag, 64, 1, MODE_GENERAL_B, 2>(p, s);
            case MODE_GENERAL_M: return launch_fwd_one<Tag, 64, 1, MODE_GENERAL_M, 2>(p, s);
            default: break;
        }
    }
#endif
    if (p.bias_f32) {   // fp32 bias next to 16-bit q / k / v (fasn_api.hip: f32_bias_vector): the fp32 image instantiations
        switch (l.mode) {
            case MODE_GENERAL: return launch_fwd_one<Tag, 64, 1, MODE_GENERAL, 2, 4, 0, 2, 0, 1, 0, 1>(p, s);
            case MODE_GENERAL_B: return launch_fwd_one<Tag, 64, 1, MODE_GENERAL_B, 2, 4, 0, 2, 0, 1, 0, 1>(p, s);
            case MODE_BIAS_KEYPAD: return launch_fwd_one<Tag, 64, 1, MODE_BIAS_KEYPAD, 2, 4, 0, 2, 0, 1, 0, 1>(p, s);
            default: break;
        }
    }
    // Round 6: the vector mask / bias modes of a large grid run EIGHT waves x 64 rows per workgroup, direct-to-LDS, one workgroup per CU (two
    // waves per SIMD, every K / V fragment read from LDS feeds two row blocks; 213 - 256 registers, no spill) - same box, (4,16,4096,64) ALiBi +
    // key padding forward: 0.443 -> 0.388 ms (-12 %; four waves x 32 rows direct-to-LDS: 0.412 / 0.425 with two / three waves per SIMD;
    // profiles/r06_d64_bias_forward_tuning_points_ab.log). Small grids keep four waves x 32 rows, register-staged.
#ifndef FASN_D64_BIAS_8WAVE
#define FASN_D64_BIAS_8WAVE 1
#endif
#ifndef FASN_D64_BIAS_8WAVE_CAUSAL
#define FASN_D64_BIAS_8WAVE_CAUSAL 2
#endif
    const long blocks512 = (long)((p.Sq + 511) / 512) * p.B * p.H;
    // (a full round of one workgroup per CU; the bias + key-padding mode pairs its batch elements by length - half as many workgroups - and needs
    // two: at (8,16,1024,64), 256 blocks, the paired 8-wave launch left half of the CUs idle, 0.089 against 0.064 ms)
    // (causal next to a mask / bias: the workgroups of a launch are unequal - a one-round launch of 512-row blocks takes as long as the same call without
    // the causal flag, the heaviest block sets the time - so the 8-wave kernel needs twice as many blocks: (4,16,2048,64) causal + bias 0.090 -> 0.068 ms, at (4,16,4096,64), two rounds, the 8-wave kernel is 2 % ahead again; profiles/r06_causal_next_to_a_bias_forward_rule_ab.log)
    if (FASN_D64_BIAS_8WAVE && blocks512 >= (l.mode == MODE_BIAS_KEYPAD ? 512 : 256) * (p.causal ? FASN_D64_BIAS_8WAVE_CAUSAL : 1) && p.Sq >= 512) {
        switch (l.mode) {
            case MODE_GENERAL: return launch_fwd_one<Tag, 64, 2, MODE_GENERAL, 2, 8, 2, 2>(p, s);
            case MODE_GENERAL_B: return launch_fwd_one<Tag, 64, 2, MODE_GENERAL_B, 2, 8, 2, 2>(p, s);
            case MODE_GENERAL_M: return launch_fwd_one<Tag, 64, 2, MODE_GENERAL_M, 2, 8, 2, 2>(p, s);
            case MODE_BIAS_KEYPAD: return launch_fwd_one<Tag, 64, 2, MODE_BIAS_KEYPAD, 2, 8, 2, 2>(p, s);
            default: break;
        }
    }
    switch (l.mode) {
        case MODE_GENERAL: return launch_fwd_one<Tag, 64, 1, MODE_GENERAL, 2, 4, 0, 2>(p, s);
        case MODE_GENERAL_B: return launch_fwd_one<Tag, 64, 1, MODE_GENERAL_B, 2, 4, 0, 2>(p, s);
        case MODE_GENERAL_M: return launch_fwd_one<Tag, 64, 1, MODE_GENERAL_M, 2, 4, 0, 2>(p, s);
        case MODE_BIAS_KEYPAD: return launch_fwd_one<Tag, 64, 1, MODE_BIAS_KEYPAD, 2, 4, 0, 2>(p, s);
        default: return launch_fwd_one<Tag, 64, 1, MODE_GENERAL_SLOW, 1>(p, s);
    }
}
#ifdef FASN_DEV_VARIANTS
template <typename Tag>
static int dev_variant(const FwdParams& p, const FwdLaunch& l, hipStream_t s) {
    switch (l.variant) {
        // ---- production tuning points
        case 100: return launch_fwd_mode<Tag, 64, 2, 2>(p, l.mode, s);
        case 90: return launch_fwd_cfg<Tag, 64, 2, 2, 4, 2, 2>(p, l.mode, s);   // the plain kernel's tuning point for any mode (causal: paired blocks by the shipped rule)
        case 91: return launch_fwd_cfg<Tag, 64, 1, 3, 4, 2, 2>(p, l.mode, s);   // the round-4 causal tuning point (32 rows per wave, three workgroups per CU) for any mode
        case 93: return launch_fwd_cfg<Tag, 64, 2, 2, 8, 2, 2>(p, l.mode, s);   // the plain kernel's tuning point with EIGHT waves per workgroup (512 rows share a K/V tile: half the L2 -> LDS traffic, one workgroup per CU)
        case 94: return launch_fwd_cfg<Tag, 64, 1, 2, 8, 2, 2>(p, l.mode, s);   // eight waves x 32 rows, two workgroups per CU
        case 96: return launch_fwd_ring<Tag, 64, 2, 2, 2, 4, 2>(p, l.mode, s);   // the plain kernel's tuning point with progress-based wave priority (PRIO 4): one-round launches
        case 97: return launch_fwd_ring<Tag, 64, 1, 3, 2, 4, 2>(p, l.mode, s);   // the 32-row tuning point with it
        case 95: if (l.mode == MODE_CAUSAL) return launch_fwd_one<Tag, 64, 2, MODE_CAUSAL, 2, 8, 2, 2, 0, 1, 1>(p, s); break;   // causal: the folded two-phase walk with eight waves (512-row blocks, rows folded w / 15 - w)
        case 92: if (l.mode == MODE_CAUSAL) return launch_fwd_one<Tag, 64, 2, MODE_CAUSAL, 2, 4, 2, 2, 0, 1, 1>(p, s); break;   // causal: folded two-phase walk whatever the launch size
        case 1: return launch_fwd_mode<Tag, 64, 1, 3>(p, l.mode, s);
        // ---- alternatives kept for A/B measurements (tools/fasn_harness bench ... <variant>)
        case 2: return launch_fwd_mode<Tag, 64, 2, 1>(p, l.mode, s);
        case 3: return launch_fwd_mode<Tag, 64, 1, 2>(p, l.mode, s);
        case 40: return launch_fwd_ring<Tag, 64, 2, 2>(p, l.mode, s);
        case 41: return launch_fwd_ring<Tag, 64, 1, 3>(p, l.mode, s);
        case 42: return launch_fwd_ring<Tag, 64, 1, 2>(p, l.mode, s);
        case 17: return launch_fwd_cfg<Tag, 64, 2, 2, 8, 1>(p, l.mode, s);   // 8 waves share a K/V tile, two-set ring
        case 18: return launch_fwd_cfg<Tag, 64, 2, 2, 8, 2>(p, l.mode, s);   // 8 waves, direct-to-LDS
        case 19: return launch_fwd_cfg<Tag, 64, 1, 3, 8, 1>(p, l.mode, s);
        case 43: return launch_fwd_ring<Tag, 64, 2, 2, 2>(p, l.mode, s);   // direct-to-LDS staging, three tile buffers
        case 44: return launch_fwd_ring<Tag, 64, 1, 3, 2>(p, l.mode, s);
        case 45: return launch_fwd_ring<Tag, 64, 1, 2, 2>(p, l.mode, s);
        case 46: return launch_fwd_ring<Tag, 64, 2, 2, 1, 2>(p, l.mode, s);   // static priority for alternate workgroups
        case 47: return launch_fwd_ring<Tag, 64, 2, 2, 1, 3>(p, l.mode, s);   // raised priority while issuing QK^T
        case 48: return launch_fwd_ring<Tag, 64, 1, 3, 1, 2>(p, l.mode, s);
        case 49: return launch_fwd_ring<Tag, 64, 1, 3, 1, 3>(p, l.mode, s);
        // seeded accumulators (pre-scaled Q, S starts at -m); with the two-set ring the 16 extra registers per row block spill
        case 82: return launch_fwd_ring<Tag, 64, 1, 2, 1, 0, 1>(p, l.mode, s);
        case 83: return launch_fwd_ring<Tag, 64, 2, 2, 2, 0, 1>(p, l.mode, s);   // + direct-to-LDS staging (no staging registers)
        case 84: return launch_fwd_ring<Tag, 64, 1, 3, 2, 0, 1>(p, l.mode, s);
        case 85: return launch_fwd_ring<Tag, 64, 2, 2, 2, 0, 2>(p, l.mode, s);   // + row sums by v_dot2c on the packed weights (= the production plain kernel)
        case 86: return launch_fwd_ring<Tag, 64, 1, 3, 2, 0, 2>(p, l.mode, s);
        case 87: return launch_fwd_ring<Tag, 64, 1, 2, 1, 0, 2>(p, l.mode, s);
        case 13: return launch_fwd_w8_mode<Tag, 64, 2, 2, 0>(p, l.mode, s); break;
        case 15: return launch_fwd_w8_mode<Tag, 64, 1, 4, 0>(p, l.mode, s); break;
        case 16: return launch_fwd_w8_mode<Tag, 64, 1, 4, 1>(p, l.mode, s); break;
        default: break;
    }
    return launch_fwd_mode<Tag, 64, 1, 3>(p, l.mode, s);
}
#endif
template <typename Tag>
static int go(const FwdParams& p, const FwdLaunch& l, hipStream_t s) {
    if (p.drop_thr) {   // dropout: the plain kernels' tuning points (round 6: the keep bits are applied to the packed weights - no spills at 64 rows per wave)
        const long bq2 = (long)((p.Sq + 255) / 256) * p.B * p.H;
        const bool big = (l.mode == MODE_PLAIN || (l.mode == MODE_KEYPAD && !p.causal)) && bq2 >= 512 && p.Sq >= 256;
        return big ? launch_fwd_drop<Tag, 64, 2, 2>(p, l.mode, s) : launch_fwd_drop<Tag, 64, 1, 3>(p, l.mode, s);
    }
    if (l.mode >= MODE_GENERAL && l.mode != MODE_KEYPAD) return launch_gen<Tag>(p, l, s);   // key-padding masks ride the plain tuning points
    // auto (what ABI callers get), measured on MI355X at (8,16,4096,64), 200 launches each. All plain / causal / key-padding
    // kernels run with seeded accumulators (Q pre-scaled, S starts at -m, row sums by v_dot2c on the packed weights):
    //   plain : QB=2 / 2 waves per SIMD / direct-to-LDS  1058 TFLOP/s  (unseeded two-set ring 1005; QB=1 / 3 waves 983); 1130 with the loop unrolled by its buffers
    //   causal: QB=1 / 3 waves per SIMD / direct-to-LDS   772 TFLOP/s  (unseeded 734; QB=2 719: coarser diagonal, worse tail)
    const long blocks_qb2 = (long)((p.Sq + 255) / 256) * p.B * p.H;
    const bool big_plain = (l.mode == MODE_PLAIN || (l.mode == MODE_KEYPAD && !p.causal)) && blocks_qb2 >= 512 && p.Sq >= 256;   // 512 = one full round of two workgroups per CU
#ifdef FASN_DEV_VARIANTS
    if (l.variant != 0) return dev_variant<Tag>(p, l, s);
#endif
    if (big_plain) return launch_fwd_cfg<Tag, 64, 2, 2, 4, 2, 2>(p, l.mode, s);
    // causal launches of at least two rounds at the plain kernel's tuning point with the folded two-phase walk (round 5; same box, final builds,
    // three alternations - profiles/r05_causal_forward_folded_two_phase_ab.log: C5 2.377 -> 2.362 ms, (16,16,4096,64) 0.6325 -> 0.6246, (4,32,8192,64)
    // 1.115 -> 1.085, (2,16,16384,64) 1.072 -> 1.017, C3 (fp16, two rounds) 0.3446 -> 0.3409; on other boxes C5 +2.1 %, C3 +0.3 .. +1.0 %)
    if (l.mode == MODE_CAUSAL && p.Sq >= 256 && blocks_qb2 >= 2048) return launch_fwd_one<Tag, 64, 2, MODE_CAUSAL, 2, 4, 2, 2, 0, 1, 1>(p, s);
    return launch_fwd_cfg<Tag, 64, 1, 3, 4, 2, 2>(p, l.mode, s);
}
int launch_fwd_d64(const FwdParams& p, const FwdLaunch& l, hipStream_t s) {
    return l.dtype == 1 ? go<bf16_tag>(p, l, s) : go<f16_tag>(p, l, s);
}
}  // namespace fasn

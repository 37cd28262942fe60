// D = 256 backward instantiations: the one-wave dQ and dK/dV kernels with the whole register file (one wave per SIMD:
// dQ^T is 128 accumulator registers next to 128 of Q / dO fragments; dK^T + dV^T would be 256 next to 128 of K / V fragments, so
// the dK/dV kernel runs as two workgroups per key block that each own half of the features, DH = 2).
// Key padding: the two-wave kernels too (kvg == 1), else the plain one-wave kernels; dense masks, 16-bit bias and dropout take the one-wave vector
// kernels (round 6), operands whose rows do not move as vectors the element-load kernels.
#include "fasn_bwd_launch.h"
#include "fasn_bwd_ws256.h"
#ifndef FASN_D256_VEC_DH
#define FASN_D256_VEC_DH 2   // feature halves of the one-wave vector dK/dV kernels (1: whole rows, 94 - 124 spilled registers; A/B in LABNOTES)
#endif
#ifndef FASN_D256_GEN_WS
#define FASN_D256_GEN_WS 1
#endif
namespace fasn {
// plain / causal without dropout or grouped K/V: the two-wave kernels of fasn_bwd_ws256.h (round 4): delta, dQ, dK/dV
// MODE = the dQ kernel's; MODE_KEYPAD (a boolean mask over (batch, head, key), p.f.causal as it comes): the dK/dV kernel then is the
// plain or causal one with the mask pointer set - it writes the rows of hidden keys as zeros (fasn_bwd_ws256.h)
template <typename Tag, int MODE>
static int launch_ws256(BwdParams p, hipStream_t s) {
    constexpr int D = 256;
    const int nbh = p.f.B * p.f.H;
    {
        constexpr int RPB = 256 / (D / 8);
        const int64_t rows = (int64_t)nbh * p.f.Sq;
        FASN_LAUNCH((fasn_bwd_delta_kernel<Tag, D>), dim3((unsigned)((rows + RPB - 1) / RPB)), dim3(256), 0, s, p);
    }
    {
        constexpr int smem = bwd_dq_ws256_smem_bytes();
        p.nblk = (p.f.Sq + 127) / 128;
        constexpr auto kern = &fasn_bwd_dq_ws256_kernel<Tag, MODE>;
        ensure_smem<kern>(smem);
        FASN_LAUNCH(kern, dim3((unsigned)(p.nblk * nbh)), dim3(512), smem, s, p);
    }
    if (p.f.kvg > 1) {   // grouped K/V (plain / causal): one workgroup per K/V head and key block walks the query heads of its group
        constexpr int smem = bwd_ws256_smem_bytes();
        p.nblk = (p.f.Sk + 127) / 128;
        if (MODE == MODE_CAUSAL) {
            constexpr auto kern = &fasn_bwd_dkdv_ws256_kernel<Tag, MODE_CAUSAL, 1>;
            ensure_smem<kern>(smem);
            FASN_LAUNCH(kern, dim3((unsigned)(p.nblk * (nbh / p.f.kvg))), dim3(512), smem, s, p);
        } else {
            constexpr auto kern = &fasn_bwd_dkdv_ws256_kernel<Tag, MODE_PLAIN, 1>;
            ensure_smem<kern>(smem);
            FASN_LAUNCH(kern, dim3((unsigned)(p.nblk * (nbh / p.f.kvg))), dim3(512), smem, s, p);
        }
    } else {
        constexpr int smem = bwd_ws256_smem_bytes();
        p.nblk = (p.f.Sk + 127) / 128;
        if (MODE == MODE_CAUSAL || (MODE == MODE_KEYPAD && p.f.causal)) {
            constexpr auto kern = &fasn_bwd_dkdv_ws256_kernel<Tag, MODE_CAUSAL>;
            ensure_smem<kern>(smem);
            FASN_LAUNCH(kern, dim3((unsigned)(p.nblk * nbh)), dim3(512), smem, s, p);
        } else {
            constexpr auto kern = &fasn_bwd_dkdv_ws256_kernel<Tag, MODE_PLAIN>;
            ensure_smem<kern>(smem);
            FASN_LAUNCH(kern, dim3((unsigned)(p.nblk * nbh)), dim3(512), smem, s, p);
        }
    }
    return launch_rc();
}
template <typename Tag>
static int go(const BwdParams& p, int mode, hipStream_t s) {
    if (mode == MODE_BIAS_KEYPAD) mode = p.f.keypad_fallback;   // bias + key padding: the dense-mask view of the same mask
    if (p.f.drop_thr) {   // dropout (round 6): ONE vector instantiation for every mode whose mask / bias rows move as vectors - no operand at all included
        const int md = mode == MODE_KEYPAD ? p.f.keypad_fallback : mode;
        if (md == MODE_GENERAL_SLOW) return launch_bwd_one<Tag, 256, 1, 1, MODE_GENERAL_SLOW, 1, 1, 1, 0, 2>(p, s);
        return launch_bwd_one<Tag, 256, 1, 1, MODE_GENERAL, 1, 1, 1, 0, FASN_D256_VEC_DH>(p, s);
    }
    if (!(FASN_BWD_VARIANT & 1)) {   // (developer library: bwd_variant bit 0 = the round-3 feature-half kernels, for A/B)
        if (mode == MODE_PLAIN) return launch_ws256<Tag, MODE_PLAIN>(p, s);      // (grouped K/V included since round 5)
        if (mode == MODE_CAUSAL) return launch_ws256<Tag, MODE_CAUSAL>(p, s);
        if (p.f.kvg == 1 && mode == MODE_KEYPAD && p.f.ms[3] == 1 && (p.f.Sk + 63) / 64 <= kDq256KpTiles) return launch_ws256<Tag, MODE_KEYPAD>(p, s);
    }
    switch (mode) {
#ifdef FASN_DEV_VARIANTS   // (plain and causal calls always take the two-wave kernels above: the one-wave causal instantiation is reachable from the A/B switch only)
        case MODE_CAUSAL: return launch_bwd_one<Tag, 256, 1, 1, MODE_CAUSAL, 1, 1, 0, 0, 2>(p, s);
#endif
        case MODE_PLAIN:   // the key-padding instantiation without a mask (every key visible): the plain one spills at this head dim
        case MODE_KEYPAD: return launch_bwd_one<Tag, 256, 1, 1, MODE_KEYPAD, 1, 1, 0, 0, 2>(p, s);
        // dense masks / 16-bit bias with vector-movable rows (round 6): the one-wave vector kernels - the dK/dV kernel with ONE additive tile
        // (two of them next to four 32 KiB Q / dO buffers would need 161 KiB of LDS; round 5 sent these calls to the element-load kernels, 3 x slower)
        case MODE_GENERAL: case MODE_GENERAL_B: case MODE_GENERAL_M:
            if (FASN_D256_GEN_WS && p.dbias == nullptr) {   // dQ by the two-wave kernel with per-wave images (no dense dS store there); dK / dV by the one-wave vector kernel
                const int nbh = p.f.B * p.f.H;
                constexpr int RPB = 256 / (256 / 8);
                const int64_t rows = (int64_t)nbh * p.f.Sq;
                FASN_LAUNCH((fasn_bwd_delta_kernel<Tag, 256>), dim3((unsigned)((rows + RPB - 1) / RPB)), dim3(256), 0, s, p);
                constexpr int smem = bwd_dq_ws256_smem_bytes(MODE_GENERAL);
                static_assert(smem <= 160 * 1024, "two-wave D = 256 dQ: LDS");
                BwdParams q = p;
                q.nblk = (p.f.Sq + 127) / 128;
                constexpr auto kern = &fasn_bwd_dq_ws256_kernel<Tag, MODE_GENERAL>;
                ensure_smem<kern>(smem);
                FASN_LAUNCH(kern, dim3((unsigned)(q.nblk * nbh)), dim3(512), smem, s, q);
                BwdParams r = p;
                r.skip |= 2 | 4;   // dQ and delta are launched
                return launch_bwd_one<Tag, 256, 1, 1, MODE_GENERAL, 1, 1, 0, 0, FASN_D256_VEC_DH>(r, s);
            }
            return launch_bwd_one<Tag, 256, 1, 1, MODE_GENERAL, 1, 1, 0, 0, FASN_D256_VEC_DH>(p, s);
        default: return launch_bwd_one<Tag, 256, 1, 1, MODE_GENERAL_SLOW, 1, 1, 0, 0, 2>(p, s);
    }
}
int launch_bwd_d256(const BwdParams& p, const FwdLaunch& l, hipStream_t s) {
    return l.dtype == 1 ? go<bf16_tag>(p, l.mode, s) : go<f16_tag>(p, l.mode, s);
}
}  // namespace fasn

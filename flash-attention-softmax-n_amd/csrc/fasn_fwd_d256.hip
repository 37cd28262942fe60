// D = 256 forward instantiations (the reference API serves any head dim through SDPA, core/flash_attn.py:117-124).
// One wave per SIMD (4-wave workgroups, 32 query rows per wave, the whole register file): Q^T fragments are 64 registers and a
// full O^T accumulator would be 128 more - the kernel spilled - so two workgroups share a query block, each with the full QK^T
// and softmax but half of the output features (VH = 2: 1.5 x the flops, no spills); K/V tiles of 64 keys (32 KiB each) are
// register-staged into two LDS buffers. Mask / bias: the vector kernel serves every combination (an absent operand is a
// zero-range descriptor / an all-ones word), key padding rides the plain kernel, everything else the element-load kernel.
#include "fasn_launch.h"
#include "fasn_fwd_ws256.h"
#ifndef FASN_D256_GEN_WS
#define FASN_D256_GEN_WS 1
#endif
namespace fasn {
// plain / causal: the two-wave kernel (fasn_fwd_ws256.h, round 4): 128-row workgroups of 8 waves, no score computed twice
template <typename Tag, int MODE, int DROP = 0>
static int launch_ws256(FwdParams p, hipStream_t s) {
    constexpr int smem = ws256_smem_bytes(MODE);
    static_assert(smem <= 160 * 1024, "two-wave D = 256 forward: LDS");
    p.nqblk = (p.Sq + 127) / 128;
    constexpr auto kern = &fasn_fwd_ws256_kernel<Tag, MODE, DROP>;
    ensure_smem<kern>(smem);
    FASN_LAUNCH(kern, dim3((unsigned)(p.nqblk * p.B * p.H)), dim3(512), smem, s, p);
    return launch_rc();
}
template <typename Tag>
static int go(const FwdParams& p, const FwdLaunch& l, hipStream_t s) {
    if (p.drop_thr) {   // dropout (round 6): the vector general instantiation whenever the call's mask / bias rows move as vectors (or there is no operand)
        const int md = (l.mode == MODE_BIAS_KEYPAD || l.mode == MODE_KEYPAD) ? p.keypad_fallback : l.mode;
        if (md == MODE_GENERAL_SLOW) return launch_fwd_one<Tag, 256, 1, MODE_GENERAL_SLOW, 1, 4, 0, 0, 1, 2>(p, s);   // the element-load kernel
        if (FASN_D256_GEN_WS) return launch_ws256<Tag, MODE_GENERAL, 1>(p, s);   // the two-wave kernel (per-wave images; keep bits on the packed weights)
        return launch_fwd_one<Tag, 256, 1, MODE_GENERAL, 1, 4, 0, 2, 1, 2>(p, s);
    }
    const int mode = l.mode == MODE_BIAS_KEYPAD ? p.keypad_fallback : l.mode;   // bias + key padding: the dense-mask view of the same mask
#ifdef FASN_DEV_VARIANTS
    if (l.variant == 1) {   // A/B: the round-3 feature-half kernels
        if (mode == MODE_PLAIN) return launch_fwd_one<Tag, 256, 1, MODE_PLAIN, 1, 4, 0, 2, 0, 2>(p, s);
        if (mode == MODE_CAUSAL) return launch_fwd_one<Tag, 256, 1, MODE_CAUSAL, 1, 4, 0, 2, 0, 2>(p, s);
        if (mode == MODE_KEYPAD) return launch_fwd_one<Tag, 256, 1, MODE_KEYPAD, 1, 4, 0, 2, 0, 2>(p, s);
    }
#endif
    switch (mode) {
        case MODE_PLAIN: return launch_ws256<Tag, MODE_PLAIN>(p, s);
        case MODE_CAUSAL: return launch_ws256<Tag, MODE_CAUSAL>(p, s);
        case MODE_KEYPAD: return launch_ws256<Tag, MODE_KEYPAD>(p, s);
        case MODE_GENERAL: case MODE_GENERAL_B: case MODE_GENERAL_M:   // dense mask and / or 16-bit bias: the two-wave kernel with per-wave images (round 6; FASN_D256_GEN_WS=0: the feature-half kernel)
            if (FASN_D256_GEN_WS) return launch_ws256<Tag, MODE_GENERAL>(p, s);
            return launch_fwd_one<Tag, 256, 1, MODE_GENERAL, 1, 4, 0, 2, 0, 2>(p, s);
        default: return launch_fwd_one<Tag, 256, 1, MODE_GENERAL_SLOW, 1, 4, 0, 0, 0, 2>(p, s);
    }
}
int launch_fwd_d256(const FwdParams& p, const FwdLaunch& l, hipStream_t s) {
    return l.dtype == 1 ? go<bf16_tag>(p, l, s) : go<f16_tag>(p, l, s);
}
}  // namespace fasn

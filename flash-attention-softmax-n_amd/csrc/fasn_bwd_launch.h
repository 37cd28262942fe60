// fasn_bwd_launch.h — host-side launch plumbing for the backward kernels.
#pragma once
#include <hip/hip_runtime.h>
#include "fasn_bwd_kernel.h"
#include "fasn_bwd_dkdv_ws.h"
#include "fasn_bwd_dq_ws.h"
#include "fasn_launch.h"

namespace fasn {

int launch_bwd(const BwdParams& p, const FwdLaunch& l, hipStream_t s);  // delta + dq + dkdv
int launch_bwd_d32(const BwdParams& p, const FwdLaunch& l, hipStream_t s);
int launch_bwd_d64(const BwdParams& p, const FwdLaunch& l, hipStream_t s);
int launch_bwd_d128(const BwdParams& p, const FwdLaunch& l, hipStream_t s);
int launch_bwd_d256(const BwdParams& p, const FwdLaunch& l, hipStream_t s);
int launch_bwd_fused_d64(const BwdParams& p, const FwdLaunch& l, hipStream_t s);   // one-pass backward (developer library only: tools/dev/fasn_bwd_fused.h), p.dqacc set
int launch_bwd_dkdv_pipe_d64(const BwdParams& p, const FwdLaunch& l, hipStream_t s);
int launch_bwd_dkdv_pipe2_d64(const BwdParams& p, const FwdLaunch& l, hipStream_t s);  // the same with one wave per SIMD and 64 keys per wave (developer A/B)
int launch_bwd_dq_pipe_d64(const BwdParams& p, const FwdLaunch& l, hipStream_t s);     // software-pipelined dQ   // software-pipelined dK/dV (fasn_bwd_pipe.h): plain / causal
int launch_bwd_dbias(const BwdParams& p, const FwdLaunch& l, int Bb, int Hb, int out_f32, hipStream_t s);   // batch- / head-reduced bias gradient (fasn_bwd_dbias.h)

// developer switch (FASN_DEV_VARIANTS builds only): bit 0 = take the one-wave dK/dV kernel where the two-wave kernel is the default,
// bit 1 = the same for dQ
#ifdef FASN_DEV_VARIANTS
extern int g_bwd_variant;
#define FASN_BWD_VARIANT g_bwd_variant
#else
#define FASN_BWD_VARIANT 0
#endif

// WS = 1: dK / dV by the two-wave kernel (fasn_bwd_dkdv_ws.h); not for dropout or the element-load mode
// BF32 = 1 (round 5): the one-wave kernels' fp32 bias instantiations (fp32 bias next to 16-bit q / k / v on the vector path; D <= 128)
template <typename Tag, int D, int QB, int KB, int MODE, int OCC_Q, int OCC_K, int DROP = 0, int WS = 0, int DH = 1, int BF32 = 0>
int launch_bwd_one(BwdParams p, hipStream_t s) {
    static_assert(!BF32 || (WS == 0 && DROP == 0 && DH == 1), "fp32 bias image: one-wave kernels without dropout");
    const int nbh = p.f.B * p.f.H;
    // delta: its own launch at head dims 128 / 256. At D <= 64 (dq_computes_delta) the dQ kernel - which runs first - computes and publishes it (round 5: row_delta in
    // fasn_bwd_kernel.h; skip bit 2 = the caller's pipelined dQ kernel does, skip bit 1 without bit 2 = the caller launches its dQ kernel AFTER the
    // dK/dV kernel of this function, a developer combination that keeps the delta launch)
    if (!(p.skip & 4) && !(dq_computes_delta(D, QB, MODE) && !(p.skip & 2))) {
        constexpr int RPB = 256 / (D / 8);
        const int64_t rows = (int64_t)nbh * p.f.Sq;
        FASN_LAUNCH((fasn_bwd_delta_kernel<Tag, D>), dim3((unsigned)((rows + RPB - 1) / RPB)), dim3(256), 0, s, p);
    }
    bool dq_done = (p.skip & 2) != 0;
    // the two-wave kernels (WS) serve every call of their modes except dropout with grouped K/V; the one-wave kernels of those modes are
    // instantiated only where they can be reached (dropout instantiations, developer A/B builds) - the plain / causal / key-padding / bias
    // one-wave dQ kernels at D = 128 spilled 11 - 15 registers and were dead code in libfasn.so
    constexpr bool WS_MODE = WS != 0 && MODE != MODE_GENERAL_SLOW && !mode_has_vmask(MODE);
#ifdef FASN_DEV_VARIANTS
    constexpr bool ONE_WAVE = true;
#else
    constexpr bool ONE_WAVE = !WS_MODE || DROP != 0;
#endif
    // a key-padding mode always fits the dQ kernel's visibility table: build_fwd hands out MODE_KEYPAD / MODE_BIAS_KEYPAD only up to the FORWARD's table
    static_assert(kFwdKpMaxTiles <= kDqWsMaxTiles, "key-padding modes: the forward's visibility table bounds the key range, the dQ kernel's must hold it");
    if constexpr (WS_MODE) {
        if (!dq_done && !(FASN_BWD_VARIANT & 2) && (!DROP || p.f.kvg == 1)) {   // dQ: two cooperating waves per row block
            constexpr int smem = 5 * KT * D * 2 + 2 * 16384 + (mode_has_vbias(MODE) ? 32768 : 0) + (mode_has_keypad(MODE) ? kDqWsMaxTiles * 8 : 0);
            p.nblk = (p.f.Sq + 127) / 128;
            constexpr auto kern = &fasn_bwd_dq_ws_kernel<Tag, D, MODE, DROP>;
            ensure_smem<kern>(smem);
            FASN_LAUNCH(kern, dim3((unsigned)(p.nblk * nbh)), dim3(512), smem, s, p);
            dq_done = true;
        }
    }
    if constexpr (ONE_WAVE) {
    if (!dq_done) {   // dQ
        constexpr int BM = 4 * QB * 32;
        constexpr int smem = 4 * KT * D * 2 + (mode_is_vector(MODE) ? 4 * QB * ((BF32 ? 8192 : 4096) + (mode_has_vmask(MODE) ? 2048 : 0)) : 0);   // + per-wave bias / mask images (the mask area only with a dense-mask operand)
        p.nblk = (p.f.Sq + BM - 1) / BM;
        constexpr auto kern = &fasn_bwd_dq_kernel<Tag, D, QB, MODE, OCC_Q, DROP, (D >= 128 ? FASN_DQ_SEED_D128 : D == 32 ? FASN_DQ_SEED_D32 : 3), BF32>;
        ensure_smem<kern>(smem);
        // causal: block r and block nblk-1-r in one workgroup (equal workgroups for the in-order dispatcher, see fasn_fwd_kernel.h)
        constexpr bool VEC_PAIR = FASN_VEC_PAIR && D <= 128 && mode_is_vector(MODE) && !mode_has_keypad(MODE);   // (the vector modes of a causal call pair their blocks too, fasn_launch.h)
        p.f.pair = ((MODE == MODE_CAUSAL || (VEC_PAIR && p.f.causal)) && (!DROP || (FASN_DROP_PAIR && MODE == MODE_CAUSAL)) && p.nblk > 1 && pair_wanted((long)p.nblk * nbh, wg_slots(OCC_Q, 4, smem), true)) ? 1 : 0;
        FASN_LAUNCH(kern, dim3((unsigned)((p.f.pair ? (p.nblk + 1) / 2 : p.nblk) * nbh)), dim3(256), smem, s, p);
        p.f.pair = 0;
    }
    }
    if (p.skip & 1) return launch_rc();
    if constexpr (WS_MODE) {
        if (!(FASN_BWD_VARIANT & 1) && (!DROP || p.f.kvg == 1)) {   // dK, dV: two cooperating waves per key block (dropout: one query head per K/V head)
            constexpr int smem = 6 * QT * D * 2 + 2 * 16384 + 6 * QT * 4 + (mode_has_vbias(MODE) ? 4 * 3 * 2048 : 0);
            p.nblk = (p.f.Sk + 127) / 128;
            if constexpr (DROP == 0) {
                if (p.f.kvg > 1) {   // grouped-query attention: one workgroup per K/V head walks the query heads of its group
                    constexpr auto kern = &fasn_bwd_dkdv_ws_kernel<Tag, D, MODE, 1>;
                    ensure_smem<kern>(smem);
                    FASN_LAUNCH(kern, dim3((unsigned)(p.nblk * (nbh / p.f.kvg))), dim3(512), smem, s, p);
                    return launch_rc();
                }
            }
            {
                constexpr auto kern = &fasn_bwd_dkdv_ws_kernel<Tag, D, MODE, 0, DROP>;
                ensure_smem<kern>(smem);
                FASN_LAUNCH(kern, dim3((unsigned)(p.nblk * nbh)), dim3(512), smem, s, p);
            }
            return launch_rc();
        }
    }
    if constexpr (ONE_WAVE) {   // dK, dV
        constexpr int BN = 4 * KB * 32;
        constexpr int smem = 4 * QT * D * 2 + 4 * QT * 4 + (mode_is_vector(MODE) ? (D == 256 ? 1 : 2) * QT * BN * (BF32 ? 4 : 2) : 0);   // (D = 256: one additive tile, see the kernel)
        static_assert(smem <= 160 * 1024, "one-wave dK/dV: LDS");
        p.nblk = (p.f.Sk + BN - 1) / BN;
        if (p.f.kvg > 1) {
            constexpr auto kern = &fasn_bwd_dkdv_kernel<Tag, D, KB, MODE, OCC_K, DROP, 1, DH, BF32>;
            ensure_smem<kern>(smem);
            FASN_LAUNCH(kern, dim3((unsigned)(p.nblk * (nbh / p.f.kvg) * DH)), dim3(256), smem, s, p);
        } else {
            constexpr auto kern = &fasn_bwd_dkdv_kernel<Tag, D, KB, MODE, OCC_K, DROP, 0, DH, BF32>;
            ensure_smem<kern>(smem);
            constexpr bool VEC_PAIRK = FASN_VEC_PAIR && mode_is_vector(MODE) && !mode_has_keypad(MODE);
            p.f.pair = ((MODE == MODE_CAUSAL || (VEC_PAIRK && p.f.causal)) && (!DROP || (FASN_DROP_PAIR && MODE == MODE_CAUSAL)) && DH == 1 && p.nblk > 1 && pair_wanted((long)p.nblk * nbh, wg_slots(OCC_K, 4, smem), true)) ? 1 : 0;
            FASN_LAUNCH(kern, dim3((unsigned)((p.f.pair ? (p.nblk + 1) / 2 : p.nblk) * nbh * DH)), dim3(256), smem, s, p);
        }
    }
    return launch_rc();
}

#ifndef FASN_DROP_BK
#define FASN_DROP_BK 1
#endif
template <typename Tag, int D, int QB, int KB, int OCC_Q, int OCC_K, int WS = 0>
int launch_bwd_mode(const BwdParams& p, int mode, hipStream_t s) {
    if (p.f.drop_thr) {   // dropout: separate instantiations (the keep-bit hash costs registers the p = 0 kernels keep)
        // the two-wave kernels (WS: D = 128) have dropout instantiations since round 4, the vector bias + key-padding mode included;
        // elsewhere that mode takes the dense-mask general mode of the same mask
        if (mode == MODE_BIAS_KEYPAD) {
            if constexpr (WS != 0) {
                if (p.f.kvg == 1) return launch_bwd_one<Tag, D, QB, KB, MODE_BIAS_KEYPAD, 1, 1, 1, WS>(p, s);
            }
            // head dims 32 / 64 (round 6): the mode's own one-wave instantiations with dropout at the vector modes' tuning point (two waves per SIMD,
            // D = 32: one block per wave) instead of the dense-mask general mode
            if constexpr (D <= 64) {
                if (FASN_DROP_BK) return launch_bwd_one<Tag, D, (D == 32 ? 1 : QB), (D == 32 ? 1 : KB), MODE_BIAS_KEYPAD, 2, 2, 1>(p, s);
            }
            mode = p.f.keypad_fallback;
        }
        if (mode == MODE_GENERAL_B) {
            if constexpr (WS != 0) {
                if (p.f.kvg == 1) return launch_bwd_one<Tag, D, QB, KB, MODE_BIAS_KEYPAD, 1, 1, 1, WS>(p, s);
            }
        }
        // D = 32: ONE 32-key block per wave in the dropout dK/dV kernels (round 6: with two, the causal / key-padding / grouped-K/V dropout
        // instantiations spilled 1 - 72 registers at their 256; with one they need 149 - 167, no spill, and three waves per SIMD fit)
        constexpr int KBD = D == 32 ? 1 : KB;
        switch (mode) {
            case MODE_PLAIN: return launch_bwd_one<Tag, D, QB, KBD, MODE_PLAIN, OCC_Q, OCC_K, 1, WS>(p, s);
            case MODE_CAUSAL: return launch_bwd_one<Tag, D, QB, KBD, MODE_CAUSAL, OCC_Q, OCC_K, 1, WS>(p, s);
            case MODE_KEYPAD: return launch_bwd_one<Tag, D, QB, KBD, MODE_KEYPAD, OCC_Q, OCC_K, 1, WS>(p, s);
            case MODE_GENERAL_SLOW: return launch_bwd_one<Tag, D, QB, KB, MODE_GENERAL_SLOW, 1, 1, 1>(p, s);
            default: return launch_bwd_one<Tag, D, (D == 32 ? 1 : QB), KBD, MODE_GENERAL, (D <= 64 ? 2 : 1), (D <= 64 ? 2 : 1), 1>(p, s);   // vector mask / bias (head dim 32: one block per wave, two waves per SIMD, as without dropout)
        }
    }
    if constexpr (D <= 128) {   // fp32 bias next to 16-bit q / k / v on the vector path (fasn_api.hip: f32_bias_vector); D = 128: the ONE-wave kernels (the two-wave ones have no LDS left for 8 KiB images)
        // Only when the call as a whole is on the vector path: build_fwd leaves bias_vec set for an aligned fp32 bias even when the MODE ends up
        // MODE_GENERAL_SLOW for another reason (a dense mask whose rows are not 4-byte movable, fp16 with |scale * log2e| > 8) - those calls must take
        // the element-load kernels below, as the forward's launch_gen does.
        if (p.f.bias_f32 && p.f.bias_vec && mode != MODE_GENERAL_SLOW) {
            if (mode == MODE_BIAS_KEYPAD) return launch_bwd_one<Tag, D, QB, KB, MODE_BIAS_KEYPAD, (D == 64 ? 2 : 1), 1, 0, 0, 1, 1>(p, s);
            return launch_bwd_one<Tag, D, QB, KB, MODE_GENERAL, (D == 64 ? 2 : 1), 1, 0, 0, 1, 1>(p, s);   // bias alone or bias + dense mask
        }
    }
    // Vector mask / bias modes on the one-wave kernels: two waves per SIMD at head dims 32 and 64. Head dim 32 (round 6): ONE 32-row / 32-key block
    // per wave - with two the kernels needed 272 / 344 registers and were compiled for one wave per SIMD; with one they need 144 / 161
    // (profiles/r06_d32_vector_modes_two_waves_per_simd_ab.log).
    constexpr int VQB = D == 32 ? 1 : QB, VKB = D == 32 ? 1 : KB, VOCC = D <= 64 ? 2 : 1;
    switch (mode) {
        case MODE_PLAIN: return launch_bwd_one<Tag, D, QB, KB, MODE_PLAIN, OCC_Q, OCC_K, 0, WS>(p, s);
        case MODE_CAUSAL: return launch_bwd_one<Tag, D, QB, KB, MODE_CAUSAL, OCC_Q, OCC_K, 0, WS>(p, s);
        case MODE_KEYPAD: return launch_bwd_one<Tag, D, QB, KB, MODE_KEYPAD, OCC_Q, OCC_K, 0, WS>(p, s);   // key-padding mask: plain kernels + visibility bits
        case MODE_GENERAL_SLOW: return launch_bwd_one<Tag, D, QB, KB, MODE_GENERAL_SLOW, 1, 1>(p, s);
        // vector bias + visibility bits (dQ at D = 128 with two waves per SIMD spills and its 88 KiB of LDS admit one workgroup per CU anyway: 12.7 vs 8.4 ms)
        case MODE_BIAS_KEYPAD: return launch_bwd_one<Tag, D, VQB, VKB, MODE_BIAS_KEYPAD, VOCC, VOCC, 0, WS>(p, s);
        case MODE_GENERAL_B:   // bias only: with the two-wave dK/dV kernel the same instantiation without a mask (every key kept)
            if constexpr (WS != 0) return launch_bwd_one<Tag, D, VQB, VKB, MODE_BIAS_KEYPAD, VOCC, VOCC, 0, WS>(p, s);
            else return launch_bwd_one<Tag, D, VQB, VKB, MODE_GENERAL, VOCC, VOCC>(p, s);
        default: return launch_bwd_one<Tag, D, VQB, VKB, MODE_GENERAL, VOCC, VOCC, 0, WS>(p, s);   // vector path (bias and/or mask)
    }
}

}  // namespace fasn

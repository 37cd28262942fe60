// fasn_launch.h — host-side launch plumbing shared by the per-head-dim translation units.
// The production launchers come first; everything under FASN_DEV_VARIANTS (A/B tuning points of the same kernels) is compiled
// only into the developer library tools/libfasn_dev.so that tools/fasn_harness links — libfasn.so carries none of it and has
// no way to select it.
#pragma once
#include <hip/hip_runtime.h>
#include <atomic>
#include <typeinfo>
#include "fasn_fwd_kernel.h"

namespace fasn {

struct FwdLaunch {
    int dtype;    // FASN_DTYPE_*
    int D;
    int mode;     // MODE_*
    int variant;  // 0 = default: the only value libfasn.so passes (FASN_DEV_VARIANTS builds take others from the harness)
};

int launch_fwd_d32(const FwdParams& p, const FwdLaunch& l, hipStream_t s);
int launch_fwd_d64(const FwdParams& p, const FwdLaunch& l, hipStream_t s);
int launch_fwd_d128(const FwdParams& p, const FwdLaunch& l, hipStream_t s);
int launch_fwd_d256(const FwdParams& p, const FwdLaunch& l, hipStream_t s);
int launch_fwd_splitk(const FwdParams& p, const FwdLaunch& l, hipStream_t s);   // p.nsplit > 1, partial buffers set

// A kernel that needs more than 48 KiB of dynamic LDS must be told so once per (kernel, device). The attribute is per
// device, so the "done" state is one bit per device ordinal of THIS kernel (the template parameter is the kernel itself:
// one static per instantiation); after the first launch on a device the launch path only reads an atomic.
// Launch recorder (fasn_launch_plan, include/fasn.h): while the calling thread has a log installed, every launch site of the library
// (FASN_LAUNCH) writes "kernel<template arguments> grid=G block=T lds=L" into it INSTEAD of launching, and no HIP call is made - the
// host side of fasn_fwd / fasn_bwd then runs to its end exactly as for a real call, so the record is the launch table itself, not a
// description of it (bench.py prints it as roofline.kernels; tests/test_spill_gate.py looks the names up in the code objects).
struct LaunchLog {
    char* buf;
    size_t cap, len;
};
extern thread_local LaunchLog* t_launch_log;
void log_launch(const char* pretty, unsigned grid, unsigned block, int smem);
template <auto Kern>
struct KernelTag {};
template <auto Kern>
const char* kernel_pretty_name() { return typeid(KernelTag<Kern>).name(); }   // mangled "fasn::KernelTag<&(void fasn::kernel<arguments>(Params))>": log_launch demangles it (__PRETTY_FUNCTION__ drops a function template's arguments)
#define FASN_LAUNCH(kern, grid, block, smem, stream, ...)                                                                    \
    do {                                                                                                                       \
        if (::fasn::t_launch_log != nullptr) ::fasn::log_launch(::fasn::kernel_pretty_name<(kern)>(), (grid).x, (block).x, (int)(smem)); \
        else hipLaunchKernelGGL(kern, grid, block, smem, stream, __VA_ARGS__);                                                \
    } while (0)

template <auto Kern>
inline void ensure_smem(int smem) {
    if (smem <= 48 * 1024 || t_launch_log != nullptr) return;
    static std::atomic<uint64_t> done{0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return;
    const uint64_t bit = 1ull << (dev & 63);
    if (dev < 64 && (done.load(std::memory_order_relaxed) & bit)) return;
    (void)hipFuncSetAttribute((const void*)Kern, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    if (dev < 64) done.fetch_or(bit, std::memory_order_relaxed);
}
inline int launch_rc() { return (t_launch_log != nullptr || hipGetLastError() == hipSuccess) ? 0 : -6; }

constexpr bool mode_is_vec(int MODE) { return mode_is_vector(MODE); }
constexpr int fwd_smem(int D, int RING, int MODE, int NW, int QB, int BF32 = 0) {
    // K/V buffers + per-wave bias / mask images (fp32 bias: 8 KiB instead of 4; the mask area only in the modes with a dense-mask operand) + key-padding visibility words
    return (RING == 2 ? 6 : 4) * KT * D * 2 + (mode_is_vec(MODE) ? NW * QB * ((BF32 ? 8192 : 4096) + (mode_has_vmask(MODE) ? 2048 : 0)) : 0) +
           (mode_has_keypad(MODE) ? kFwdKpMaxTiles * 8 : 0);
}

// one instantiation of the forward kernel: NW waves x QB 32-row blocks per wave, staging scheme RING, accumulator seeding SEED
// Paired causal blocks (block r and block nblk-1-r in one workgroup: equal workgroups, see the kernels). Rounds 2 - 5: from two rounds of single blocks.
// Round 6 (profiles/r06_causal_pairing_threshold_ab.log, (4,16,S,D) causal): the in-order dispatcher hands unequal workgroups out head by head, so the heavy
// blocks of the last heads start late, and even a launch that is resident at once puts two heavy blocks on one CU - (4,16,2048,64) forward 0.064 -> 0.054 ms
// paired at 1.33 rounds, (4,16,2048,32) -13 % at exactly one round, (4,16,2048,128) 0.117 -> 0.090 (its 96 KiB of LDS make 512 blocks TWO rounds: `slots`
// now counts LDS). But pairs that do not fill the CUs evenly cost the forward more than they save below two rounds ((4,16,1536,64): 384 pairs on 256 CUs,
// +7 % plain, +21 % with a bias), while dQ / dK/dV gained in every case measured from 1.25 rounds on. So: two rounds, or - forward - one round and a multiple
// of 256 pairs, or - backward - 1.25 rounds.
constexpr long wg_slots(int occ_waves_per_simd, int nw, int smem) {   // workgroups the chip holds at once
    const long by_regs = occ_waves_per_simd * 4 / nw < 1 ? 1 : occ_waves_per_simd * 4 / nw;
    const long by_lds = smem > 0 ? 163840 / smem : by_regs;
    return 256L * (by_regs < by_lds ? by_regs : (by_lds < 1 ? 1 : by_lds));
}
#ifndef FASN_PAIR_RULE
#define FASN_PAIR_RULE 6   // 5: the rule of rounds 2 - 5 (two rounds of single blocks; slots by registers only)
#endif
inline bool pair_rule(long blocks, long slots, bool backward) {
    if (blocks >= 2 * slots) return true;
    if (FASN_PAIR_RULE == 5) return false;
    if (backward) return 4 * blocks >= 5 * slots;
    return blocks >= slots && blocks % 512 == 0;
}
#ifdef FASN_DEV_VARIANTS
extern int g_kprot;       // developer library: 0 = no rotated second pass of a length pair (A/B)
extern int* g_xq;         // developer experiment: item counters of the dynamic deal across XCDs (zeroed here before every launch)
extern int g_xq_extra;
extern int g_pair_mode;   // developer library: -1 = shipped rule, 0 = never pair, 1 = always pair (tools/fasn_harness, env FASN_PAIR)
inline bool pair_wanted(long blocks, long slots, bool backward = false) { return g_pair_mode < 0 ? pair_rule(blocks, slots, backward) : g_pair_mode != 0; }
#else
inline bool pair_wanted(long blocks, long slots, bool backward = false) { return pair_rule(blocks, slots, backward); }
#endif
template <typename Tag, int D, int QB, int MODE, int OCC, int NW = 4, int RING = 0, int SEED = 0, int DROP = 0, int VH = 1, int FOLD = 0, int BF32 = 0>
int launch_fwd_one(FwdParams p, hipStream_t s) {
    constexpr int BM = NW * QB * 32;
    constexpr int smem = fwd_smem(D, RING, MODE, NW, QB, BF32);
    p.nqblk = (p.Sq + BM - 1) / BM;
    constexpr auto kern = &fasn_fwd_kernel<Tag, D, QB, MODE, OCC, NW, (NW == 8 ? FASN_PRIO8 : 0), DROP, RING, 0, SEED, VH, FOLD, BF32>;
    ensure_smem<kern>(smem);
    // causal: pair block r with block nqblk-1-r in one workgroup (equal workgroups, see the kernel) when the single blocks fill the
    // chip's workgroup slots at least kPairRounds times; smaller launches keep single blocks, heaviest first
    int blocks = p.nqblk;
    p.pair = 0;
    // (round 6: also the vector mask / bias modes when the call is causal - ALiBi in a decoder: without pairs a launch of unequal workgroups handed out
    // head by head ends on heavy blocks that started late; (4,16,2048,64) causal + bias ran at 0.84 of the non-causal time instead of ~0.55)
    constexpr bool VEC_PAIR = FASN_VEC_PAIR && mode_is_vector(MODE) && !mode_has_keypad(MODE);
    if ((MODE == MODE_CAUSAL || (VEC_PAIR && p.causal)) && VH == 1 && (!DROP || (FASN_DROP_PAIR && MODE == MODE_CAUSAL)) && p.nqblk > 1 && pair_wanted((long)p.nqblk * p.B * p.H, wg_slots(OCC, NW, smem))) {
        p.pair = 1;
        blocks = (p.nqblk + 1) / 2;
    }
#ifdef FASN_DEV_VARIANTS
    // length-paired batch elements (kpair_plan): developer override through the same switch - 2 = never, 3 = whatever the lengths
    if (mode_has_keypad(MODE) && mode_has_vbias(MODE) && g_pair_mode >= 0) p.pair = g_pair_mode ? 3 : 2;
#endif
    if constexpr (fwd_xq_kernel(D, MODE, 0, VH, DROP)) {
        if (p.xq != nullptr && ((p.B * p.H) & 7) == 0) {   // dynamic deal of the items across XCDs (fasn_fwd_ws; the counters are the caller's workspace, zeroed here)
            int surplus = kXqSurplus;
#ifdef FASN_DEV_VARIANTS
            if (g_xq != nullptr) surplus = g_xq_extra;
#endif
            // counters that could not be zeroed (bad workspace pointer, a capture that refuses the node) would make workgroups skip or repeat
            // items: such a launch takes the static deal below instead
            const bool zeroed = t_launch_log != nullptr || hipMemsetAsync(p.xq, 0, 8 * sizeof(int), s) == hipSuccess;
            if (zeroed) {
                FASN_LAUNCH(kern, dim3((unsigned)(blocks * p.B * p.H + 8 * surplus)), dim3(NW * 64), smem, s, p);
                return launch_rc();
            }
            (void)hipGetLastError();
        }
    }
    p.xq = nullptr;   // (every other instantiation: static deal)
    FASN_LAUNCH(kern, dim3((unsigned)(blocks * p.B * p.H * VH)), dim3(NW * 64), smem, s, p);
    return launch_rc();
}

// plain / causal / key-padding launch of one (workgroup size, staging scheme) tuning point
template <typename Tag, int D, int QB, int OCC, int NW, int RING, int SEED = 0>
int launch_fwd_cfg(const FwdParams& p, int mode, hipStream_t s) {
    if (mode == MODE_PLAIN) return launch_fwd_one<Tag, D, QB, MODE_PLAIN, OCC, NW, RING, SEED>(p, s);
    if (mode == MODE_KEYPAD) return launch_fwd_one<Tag, D, QB, MODE_KEYPAD, OCC, NW, RING, SEED>(p, s);
    return launch_fwd_one<Tag, D, QB, MODE_CAUSAL, OCC, NW, RING, SEED>(p, s);
}

// dropout instantiations: plain, causal, key-padding, the vector mask / bias kernel (MODE_GENERAL serves all three mask / bias
// combinations: an absent operand is a zero-range descriptor / an all-ones word) and the element-load general kernel.
// Round 6 (dropout stream definition 2, fasn_common.h): the keep bits are applied to the PACKED weights, behind the row sums - the dropout
// kernels run at the plain kernels' tuning points, seeded accumulators and packed row sums (SEED = 2) included: 64 rows per wave at D = 64 no
// longer spill (244 registers; the round-5 hash needed 37 - 67 more), three waves per SIMD fit in 144.
// QB / OCC: the tuning point of the plain kernel for this launch (fasn_fwd_d*.hip decides by the size of the grid).
#ifndef FASN_DROP_BK
#define FASN_DROP_BK 1
#endif
template <typename Tag, int D, int QB, int OCC>
int launch_fwd_drop(const FwdParams& p, int mode, hipStream_t s) {
    if (mode == MODE_BIAS_KEYPAD) {
        // vector bias + key-padding mask (ALiBi on a padded batch) with dropout: the visibility-word kernel at D = 128 (round 4), elsewhere
        // the dense-mask general mode of the same mask
        if constexpr (D == 128) return launch_fwd_one<Tag, D, 1, MODE_BIAS_KEYPAD, 2, 8, 2, 2, 1>(p, s);
        // head dims 32 / 64 (round 6, FASN_DROP_BK): the same mode at the vector dropout kernels' tuning point (154 / 174 registers) instead of the
        // dense-mask general mode of the same mask (a mask image per tile next to the bias image)
        else if (FASN_DROP_BK) return launch_fwd_one<Tag, D, (D == 32 ? 2 : 1), MODE_BIAS_KEYPAD, 2, 4, (D == 32 ? 2 : 0), 2, 1>(p, s);
        else mode = p.keypad_fallback;
    }
    if (mode == MODE_PLAIN) {
        if constexpr (D == 128) return launch_fwd_one<Tag, D, 1, MODE_PLAIN, 2, 8, 2, 2, 1>(p, s);   // 8 waves share a K/V tile, two per SIMD
        else return launch_fwd_one<Tag, D, QB, MODE_PLAIN, OCC, 4, 2, 2, 1>(p, s);
    }
    if (mode == MODE_CAUSAL) {
        if constexpr (D == 128) return launch_fwd_one<Tag, D, 1, MODE_CAUSAL, 2, 8, 2, 2, 1>(p, s);
        else return launch_fwd_one<Tag, D, QB, MODE_CAUSAL, OCC, 4, 2, 2, 1>(p, s);
    }
    if (mode == MODE_KEYPAD) {
        if constexpr (D == 128) return launch_fwd_one<Tag, D, 1, MODE_KEYPAD, 2, 8, 2, 2, 1>(p, s);
        else return launch_fwd_one<Tag, D, QB, MODE_KEYPAD, OCC, 4, 2, 2, 1>(p, s);
    }
    if (mode == MODE_GENERAL || mode == MODE_GENERAL_B || mode == MODE_GENERAL_M) {
        if constexpr (D == 128) return launch_fwd_one<Tag, D, 1, MODE_GENERAL, 2, 8, 2, 2, 1>(p, s);
        else return launch_fwd_one<Tag, D, (D == 32 ? 2 : 1), MODE_GENERAL, 2, 4, (D == 32 ? 2 : 0), 2, 1>(p, s);   // (the p = 0 vector kernels' small-grid tuning points)
    }
    return launch_fwd_one<Tag, D, (D == 32 ? 2 : 1), MODE_GENERAL_SLOW, 1, 4, 0, 0, 1>(p, s);
}

#ifdef FASN_DEV_VARIANTS
// ------------------------------------------------------------------------------------------------------------------
// developer launchers (tools/fasn_harness bench ... <variant>)
template <typename Tag, int D, int QB, int OCC>
int launch_fwd_mode(const FwdParams& p, int mode, hipStream_t s) {   // unseeded, register-staged (the round-1 baseline)
    return launch_fwd_cfg<Tag, D, QB, OCC, 4, 0, 0>(p, mode, s);
}

// two staging register sets / direct-to-LDS with a static wave priority
template <typename Tag, int D, int QB, int MODE, int OCC, int RING = 1, int PRIO = 0, int SEED = 0, int NW = 4>
int launch_fwd_ring_one(FwdParams p, hipStream_t s) {
    constexpr int BM = NW * QB * 32;
    constexpr int smem = fwd_smem(D, RING, MODE, NW, QB);
    p.nqblk = (p.Sq + BM - 1) / BM;
    constexpr auto kern = &fasn_fwd_kernel<Tag, D, QB, MODE, OCC, NW, PRIO, 0, RING, 0, SEED>;
    ensure_smem<kern>(smem);
    FASN_LAUNCH(kern, dim3((unsigned)(p.nqblk * p.B * p.H)), dim3(NW * 64), smem, s, p);
    return launch_rc();
}
template <typename Tag, int D, int QB, int OCC, int RING = 1, int PRIO = 0, int SEED = 0>
int launch_fwd_ring(const FwdParams& p, int mode, hipStream_t s) {
    if (mode == MODE_PLAIN) return launch_fwd_ring_one<Tag, D, QB, MODE_PLAIN, OCC, RING, PRIO, SEED>(p, s);
    if (mode == MODE_KEYPAD) return launch_fwd_ring_one<Tag, D, QB, MODE_KEYPAD, OCC, RING, PRIO, SEED>(p, s);
    return launch_fwd_ring_one<Tag, D, QB, MODE_CAUSAL, OCC, RING, PRIO, SEED>(p, s);
}
template <typename Tag, int D, int QB, int OCC, int PRIO>
int launch_fwd_w8_mode(const FwdParams& p, int mode, hipStream_t s) {
    if (mode == MODE_PLAIN) return launch_fwd_ring_one<Tag, D, QB, MODE_PLAIN, OCC, 0, PRIO, 0, 8>(p, s);
    return launch_fwd_ring_one<Tag, D, QB, MODE_CAUSAL, OCC, 0, PRIO, 0, 8>(p, s);
}

#endif  // FASN_DEV_VARIANTS

}  // namespace fasn

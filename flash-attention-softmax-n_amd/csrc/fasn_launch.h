// fasn_launch.h — host-side launch plumbing shared by the per-head-dim translation units.
#pragma once
#include <hip/hip_runtime.h>
#include "fasn_fwd_kernel.h"
#include "fasn_fwd_pipe.h"
#include "fasn_fwd_pp.h"
#include "fasn_fwd_split.h"

namespace fasn {

// tuning variant (internal, not part of the C ABI): selects QB (32-row query blocks per wave)
struct FwdLaunch {
    int dtype;    // FASN_DTYPE_*
    int D;
    int mode;     // MODE_*
    int variant;  // 0 = default
};

int launch_fwd_d32(const FwdParams& p, const FwdLaunch& l, hipStream_t s);
int launch_fwd_d64(const FwdParams& p, const FwdLaunch& l, hipStream_t s);
int launch_fwd_d128(const FwdParams& p, const FwdLaunch& l, hipStream_t s);
int launch_fwd_splitk(const FwdParams& p, const FwdLaunch& l, hipStream_t s);   // p.nsplit > 1, partial buffers set


template <typename K>
inline void set_smem_attr(K kern, int smem) {
    if (smem > 48 * 1024) (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
}

template <typename Tag, int D, int QB, int MODE, int OCC, int NW = 4, int RING = 0, int SEED = 0>
int launch_fwd_one(FwdParams p, hipStream_t s) {
    constexpr int BM = NW * QB * 32;
    constexpr bool VEC = MODE == MODE_GENERAL || MODE == MODE_GENERAL_B || MODE == MODE_GENERAL_M;
    constexpr int smem = (RING == 2 ? 6 : 4) * KT * D * 2 + (VEC ? NW * QB * 6144 : 0);   // + per-wave bias / mask images
    p.nqblk = (p.Sq + BM - 1) / BM;
    auto kern = fasn_fwd_kernel<Tag, D, QB, MODE, OCC, NW, 0, 0, 0, RING, 0, SEED>;
    if (smem > 48 * 1024) {
        static bool done = false;  // benign race: idempotent attribute
        if (!done) {
            (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
            done = true;
        }
    }
    const dim3 grid((unsigned)(p.nqblk * p.B * p.H));
    hipLaunchKernelGGL(kern, grid, dim3(NW * 64), smem, s, p);
    return hipGetLastError() == hipSuccess ? 0 : -6;
}

template <typename Tag, int D, int QB, int MODE, int OCC, int BURST = 0>
int launch_fwd_pipe_one(FwdParams p, hipStream_t s) {
    constexpr int BM = 4 * QB * 32;
    constexpr int smem = 4 * KT * D * 2;
    p.nqblk = (p.Sq + BM - 1) / BM;
    auto kern = fasn_fwd_pipe_kernel<Tag, D, QB, MODE, OCC, BURST>;
    if (smem > 48 * 1024) {
        static bool done = false;
        if (!done) {
            (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
            done = true;
        }
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)(p.nqblk * p.B * p.H)), dim3(256), smem, s, p);
    return hipGetLastError() == hipSuccess ? 0 : -6;
}

// developer ablation launcher (plain mode, 4 waves)
template <typename Tag, int D, int QB, int OCC, int ABL>
int launch_fwd_abl(FwdParams p, hipStream_t s) {
    constexpr int BM = 4 * QB * 32;
    constexpr int smem = 4 * KT * D * 2;
    p.nqblk = (p.Sq + BM - 1) / BM;
    auto kern = fasn_fwd_kernel<Tag, D, QB, MODE_PLAIN, OCC, 4, 0, ABL>;
    hipLaunchKernelGGL(kern, dim3((unsigned)(p.nqblk * p.B * p.H)), dim3(256), smem, s, p);
    return hipGetLastError() == hipSuccess ? 0 : -6;
}

// plain / causal kernel with two staging register sets (K/V loaded two tiles ahead)
template <typename Tag, int D, int QB, int MODE, int OCC, int RING = 1, int PRIO = 0, int SEED = 0>
int launch_fwd_ring_one(FwdParams p, hipStream_t s) {
    constexpr int BM = 4 * QB * 32;
    constexpr int smem = (RING == 2 ? 6 : 4) * KT * D * 2;
    p.nqblk = (p.Sq + BM - 1) / BM;
    auto kern = fasn_fwd_kernel<Tag, D, QB, MODE, OCC, 4, PRIO, 0, 0, RING, 0, SEED>;
    if (smem > 48 * 1024) (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    hipLaunchKernelGGL(kern, dim3((unsigned)(p.nqblk * p.B * p.H)), dim3(256), smem, s, p);
    return hipGetLastError() == hipSuccess ? 0 : -6;
}
template <typename Tag, int D, int QB, int OCC, int RING = 1, int PRIO = 0, int SEED = 0>
int launch_fwd_ring(const FwdParams& p, int mode, hipStream_t s) {
    if (mode == MODE_PLAIN) return launch_fwd_ring_one<Tag, D, QB, MODE_PLAIN, OCC, RING, PRIO, SEED>(p, s);
    if (mode == MODE_KEYPAD) return launch_fwd_ring_one<Tag, D, QB, MODE_KEYPAD, OCC, RING, PRIO, SEED>(p, s);
    return launch_fwd_ring_one<Tag, D, QB, MODE_CAUSAL, OCC, RING, PRIO, SEED>(p, s);
}

// key-block-split kernel (fasn_fwd_split.h)
template <typename Tag, int D, int QB, int MODE, int OCC>
int launch_fwd_split_one(FwdParams p, hipStream_t s) {
    constexpr int BM = 4 * QB * 32;
    constexpr int smem = 4 * KT * D * 2;
    p.nqblk = (p.Sq + BM - 1) / BM;
    auto kern = fasn_fwd_split_kernel<Tag, D, QB, MODE, OCC>;
    if (smem > 48 * 1024) (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    hipLaunchKernelGGL(kern, dim3((unsigned)(p.nqblk * p.B * p.H)), dim3(256), smem, s, p);
    return hipGetLastError() == hipSuccess ? 0 : -6;
}
template <typename Tag, int D, int QB, int OCC>
int launch_fwd_split(const FwdParams& p, int mode, hipStream_t s) {
    if (mode == MODE_PLAIN) return launch_fwd_split_one<Tag, D, QB, MODE_PLAIN, OCC>(p, s);
    return launch_fwd_split_one<Tag, D, QB, MODE_CAUSAL, OCC>(p, s);
}

// 8-wave workgroups of the plain kernel (QB 32-row blocks per wave), optional static priority for waves 4-7
template <typename Tag, int D, int QB, int MODE, int OCC, int PRIO>
int launch_fwd_w8_one(FwdParams p, hipStream_t s) {
    constexpr int BM = 8 * QB * 32;
    constexpr int smem = 4 * KT * D * 2;
    p.nqblk = (p.Sq + BM - 1) / BM;
    auto kern = fasn_fwd_kernel<Tag, D, QB, MODE, OCC, 8, PRIO>;
    if (smem > 48 * 1024) {
        static bool done = false;
        if (!done) {
            (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
            done = true;
        }
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)(p.nqblk * p.B * p.H)), dim3(512), smem, s, p);
    return hipGetLastError() == hipSuccess ? 0 : -6;
}
template <typename Tag, int D, int QB, int OCC, int PRIO>
int launch_fwd_w8_mode(const FwdParams& p, int mode, hipStream_t s) {
    if (mode == MODE_PLAIN) return launch_fwd_w8_one<Tag, D, QB, MODE_PLAIN, OCC, PRIO>(p, s);
    return launch_fwd_w8_one<Tag, D, QB, MODE_CAUSAL, OCC, PRIO>(p, s);
}

template <typename Tag, int D, int MODE, int OCC>
int launch_fwd_pp_one(FwdParams p, hipStream_t s) {
    constexpr int BM = 256;
    constexpr int smem = 4 * KT * D * 2;
    p.nqblk = (p.Sq + BM - 1) / BM;
    auto kern = fasn_fwd_pp_kernel<Tag, D, MODE, OCC>;
    if (smem > 48 * 1024) {
        static bool done = false;
        if (!done) {
            (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
            done = true;
        }
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)(p.nqblk * p.B * p.H)), dim3(512), smem, s, p);
    return hipGetLastError() == hipSuccess ? 0 : -6;
}
template <typename Tag, int D, int OCC>
int launch_fwd_pp_mode(const FwdParams& p, int mode, hipStream_t s) {
    if (mode == MODE_PLAIN) return launch_fwd_pp_one<Tag, D, MODE_PLAIN, OCC>(p, s);
    return launch_fwd_pp_one<Tag, D, MODE_CAUSAL, OCC>(p, s);
}

// developer ablation launcher for 8-wave workgroups (plain mode)
template <typename Tag, int D, int QB, int OCC, int ABL>
int launch_fwd_abl8(FwdParams p, hipStream_t s) {
    constexpr int BM = 8 * QB * 32;
    constexpr int smem = 4 * KT * D * 2;
    p.nqblk = (p.Sq + BM - 1) / BM;
    auto kern = fasn_fwd_kernel<Tag, D, QB, MODE_PLAIN, OCC, 8, 0, ABL>;
    set_smem_attr(kern, smem);
    hipLaunchKernelGGL(kern, dim3((unsigned)(p.nqblk * p.B * p.H)), dim3(512), smem, s, p);
    return hipGetLastError() == hipSuccess ? 0 : -6;
}

// any (workgroup size, staging scheme) combination of the plain / causal kernel
template <typename Tag, int D, int QB, int OCC, int NW, int RING, int SEED = 0>
int launch_fwd_cfg(FwdParams p, int mode, hipStream_t s) {
    constexpr int BM = NW * QB * 32;
    constexpr int smem = (RING == 2 ? 6 : 4) * KT * D * 2;
    p.nqblk = (p.Sq + BM - 1) / BM;
    const dim3 grid((unsigned)(p.nqblk * p.B * p.H)), block(NW * 64);
    if (mode == MODE_PLAIN) {
        auto kern = fasn_fwd_kernel<Tag, D, QB, MODE_PLAIN, OCC, NW, 0, 0, 0, RING, 0, SEED>;
        set_smem_attr(kern, smem);
        hipLaunchKernelGGL(kern, grid, block, smem, s, p);
    } else if (mode == MODE_KEYPAD) {
        auto kern = fasn_fwd_kernel<Tag, D, QB, MODE_KEYPAD, OCC, NW, 0, 0, 0, RING, 0, SEED>;
        set_smem_attr(kern, smem);
        hipLaunchKernelGGL(kern, grid, block, smem, s, p);
    } else {
        auto kern = fasn_fwd_kernel<Tag, D, QB, MODE_CAUSAL, OCC, NW, 0, 0, 0, RING, 0, SEED>;
        set_smem_attr(kern, smem);
        hipLaunchKernelGGL(kern, grid, block, smem, s, p);
    }
    return hipGetLastError() == hipSuccess ? 0 : -6;
}

// pipelined kernel for plain / causal; the general (mask / bias) mode stays on fasn_fwd_kernel
template <typename Tag, int D, int QB, int OCC, int BURST = 0>
int launch_fwd_pipe_mode(const FwdParams& p, int mode, hipStream_t s) {
    if (mode == MODE_PLAIN) return launch_fwd_pipe_one<Tag, D, QB, MODE_PLAIN, OCC, BURST>(p, s);
    return launch_fwd_pipe_one<Tag, D, QB, MODE_CAUSAL, OCC, BURST>(p, s);
}

// dropout instantiations: plain, causal, and the element-load general kernel (any mask / bias combination)
template <typename Tag, int D, int QB, int MODE, int OCC>
int launch_fwd_drop_one(FwdParams p, hipStream_t s) {
    constexpr int BM = 4 * QB * 32;
    constexpr int smem = 4 * KT * D * 2;
    p.nqblk = (p.Sq + BM - 1) / BM;
    constexpr int SEED = MODE == MODE_GENERAL_SLOW ? 0 : 1;   // seeded S accumulators; row sums stay fp32 (taken before the drop)
    auto kern = fasn_fwd_kernel<Tag, D, QB, MODE, OCC, 4, 0, 0, 1, 0, 0, SEED>;
    if (smem > 48 * 1024) (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    hipLaunchKernelGGL(kern, dim3((unsigned)(p.nqblk * p.B * p.H)), dim3(256), smem, s, p);
    return hipGetLastError() == hipSuccess ? 0 : -6;
}
// dropout + vector mask / bias (the usual fine-tuning setting: padding mask + attention dropout): MODE_GENERAL serves all
// three mask / bias combinations (an absent operand is a zero-range descriptor / an all-ones word)
template <typename Tag, int D, int OCC, int NW, int RING>
int launch_fwd_drop_gen(FwdParams p, hipStream_t s) {
    constexpr int BM = NW * 32;
    constexpr int smem = (RING == 2 ? 6 : 4) * KT * D * 2 + NW * 6144;
    p.nqblk = (p.Sq + BM - 1) / BM;
    auto kern = fasn_fwd_kernel<Tag, D, 1, MODE_GENERAL, OCC, NW, 0, 0, 1, RING, 0, 1>;
    set_smem_attr(kern, smem);
    hipLaunchKernelGGL(kern, dim3((unsigned)(p.nqblk * p.B * p.H)), dim3(NW * 64), smem, s, p);
    return hipGetLastError() == hipSuccess ? 0 : -6;
}
template <typename Tag, int D, int QB, int OCC>
int launch_fwd_drop(const FwdParams& p, int mode, hipStream_t s) {
    if (mode == MODE_PLAIN) return launch_fwd_drop_one<Tag, D, QB, MODE_PLAIN, OCC>(p, s);
    if (mode == MODE_CAUSAL) return launch_fwd_drop_one<Tag, D, QB, MODE_CAUSAL, OCC>(p, s);
    if (mode == MODE_KEYPAD) return launch_fwd_drop_one<Tag, D, QB, MODE_KEYPAD, OCC>(p, s);
    if (mode == MODE_GENERAL || mode == MODE_GENERAL_B || mode == MODE_GENERAL_M) {
        if constexpr (D == 128) return launch_fwd_drop_gen<Tag, D, 2, 8, 2>(p, s);
        else return launch_fwd_drop_gen<Tag, D, D == 32 ? 1 : 2, 4, 0>(p, s);
    }
    return launch_fwd_drop_one<Tag, D, QB, MODE_GENERAL_SLOW, 1>(p, s);
}

template <typename Tag, int D, int QB, int OCC>
int launch_fwd_mode(const FwdParams& p, int mode, hipStream_t s) {
    switch (mode) {
        case MODE_PLAIN: return launch_fwd_one<Tag, D, QB, MODE_PLAIN, OCC>(p, s);
        case MODE_CAUSAL: return launch_fwd_one<Tag, D, QB, MODE_CAUSAL, OCC>(p, s);
        case MODE_KEYPAD: return launch_fwd_one<Tag, D, QB, MODE_KEYPAD, OCC>(p, s);
        default: return -7;  // general (mask / bias) mode is dispatched explicitly by the per-D translation units
    }
}

}  // namespace fasn

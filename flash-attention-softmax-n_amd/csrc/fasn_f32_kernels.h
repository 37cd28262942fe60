// fasn_f32_kernels.h — fp32-in / fp32-out attention-softmax_n (forward, delta, dQ, dK/dV) on the exact-fp32 matrix
// instruction v_mfma_f32_32x32x2_f32 (64 cycles per instruction: the fp32 vector rate, 1/16 of the bf16 MFMA rate; there is
// no TF32-class fast path on gfx950). Same orientation and tile schedule as the 16-bit kernels (fasn_fwd_kernel.h,
// fasn_bwd_kernel.h): a lane owns a query row (forward, dQ) or a key column (dK/dV); what differs is the operand plumbing:
//   * operand with the contraction along the feature dim: lane reads 16 B = 4 consecutive floats of its row (ds_read_b128 /
//     global load); MFMA e of that chunk pairs lane-half 0's float e (feature 8c+e) with lane-half 1's (feature 8c+4+e)
//   * operand with the contraction along tile rows (V^T, K^T, dO^T, Q^T): one ds_read_b32 per MFMA, row = the row that
//     accumulator register r of this lane-half stands for, column = lane&31 — no transpose instruction needed
//   * P / dS stay fp32: accumulator register r IS the B operand of MFMA r (no packing, no rounding)
// Modes: plain and causal (ragged sizes handled); masks, bias and dropout are 16-bit-path features.
// Reference: flash_attention_softmax_n/core/flash_attn.py:42-124 and tests/gpu/core/test_flash_attn.py:14 (fp32 atol 1e-3).
#pragma once
#include "fasn_bwd_kernel.h"

namespace fasn {

FASN_DEV f32x16 mfma32(float a, float b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0); }

// 16-byte-chunk swizzle of a [rows][D] fp32 tile: conflict-free for "row = lane&31, same chunk" ds_read_b128
template <int D>
FASN_DEV int swz32(int row) {
    if constexpr (D == 32) return (row >> 1) & 7;  // 128-B rows, two per bank row
    else return row & 15;                          // 256-B / 512-B rows
}
template <int D>
FASN_DEV int off32(int row, int chunk) { return row * (D * 4) + ((chunk ^ swz32<D>(row)) << 4); }
// scalar element (row, col)
template <int D>
FASN_DEV float lds_elem32(const char* tile, int row, int col) {
    return *LDS_PTR(const float, tile + off32<D>(row, col >> 2) + (col & 3) * 4);
}
template <int D>
FASN_DEV f32x4 lds_row4(const char* tile, int row, int chunk) { return *LDS_PTR(const f32x4, tile + off32<D>(row, chunk)); }

// stage a [64][D] fp32 tile: buffer loads (rows past the end read as 0) -> registers -> LDS
template <int D>
struct Stage32 {
    static constexpr int CPR = D / 4;                 // 16-byte chunks per row
    static constexpr int NLD = (64 * CPR) / 256;
    static constexpr int RPI = 256 / CPR;             // rows between two pieces of a thread
    // piece i of a thread is row (tid / CPR) + RPI * i, same chunk: its global offset differs by a wave-uniform amount (added on the scalar
    // side) and its LDS offset by a constant - except at D = 128 (RPI = 8, the swizzle is row & 15), where odd pieces flip bit 3 of the chunk
    unsigned voff;
    int loff[2];
    FASN_DEV void init(int tid, int64_t row_stride) {
        const int row = tid / CPR, ch = tid % CPR;
        voff = (unsigned)(row * (int)row_stride * 4 + ch * 16);
        loff[0] = off32<D>(row, ch);
        loff[1] = off32<D>(row + RPI, ch) - RPI * D * 4;
    }
    FASN_DEV void gload(u32x4 (&st)[NLD], __amdgpu_buffer_rsrc_t rs, int row0, int64_t row_stride) const {
#pragma unroll
        for (int i = 0; i < NLD; ++i) st[i] = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, (row0 + RPI * i) * (int)row_stride * 4, 0);
    }
    FASN_DEV void lstore(const u32x4 (&st)[NLD], char* tile) const {
#pragma unroll
        for (int i = 0; i < NLD; ++i) *LDS_PTR(u32x4, tile + loff[D == 128 ? (i & 1) : 0] + i * (RPI * D * 4)) = st[i];
    }
};

// row of the accumulator register r for lane-half hi
FASN_DEV int acc_row(int r, int hi) { return (r & 3) + 8 * (r >> 2) + 4 * hi; }

// MODE_GENERAL_SLOW in the fp32 kernels = mask and/or bias and/or dropout, element loads through the four strides (the fp32
// kernels run at the fp32 MFMA rate, 1/16 of bf16, so the loads are not what limits them). Bias elements are fp32 here.
struct GenElem {
    const float* bias;
    const uint8_t* mask;
    FASN_DEV void init(const FwdParams& p, int b, int h) {
        bias = p.bias ? reinterpret_cast<const float*>(p.bias) + (b * p.bs[0] + h * p.bs[1]) : nullptr;
        mask = p.mask ? p.mask + (b * p.ms[0] + h * p.ms[1]) : nullptr;
    }
    // y (log2 domain) += bias*log2e; returns false when the mask hides the element. (row, key) must be in range.
    FASN_DEV bool apply(const FwdParams& p, int row, int key, float& y) const {
        if (bias) y = __builtin_fmaf(bias[(int64_t)row * p.bs[2] + (int64_t)key * p.bs[3]], kLog2e, y);
        return mask ? mask[(int64_t)row * p.ms[2] + (int64_t)key * p.ms[3]] != 0 : true;
    }
};
// dropout of the fp32 kernels: the launch's seed pair and thresholds, read once per kernel; one weight at a time (fasn_common.h: drop_keep_at)
struct F32Drop {
    DropSeed dsd;
    uint32_t thr;
    FASN_DEV explicit F32Drop(const FwdParams& p) : dsd(p.drop_thr ? drop_seed(p.seed_lo, p.seed_hi, p.rng) : DropSeed{0u, 0u}), thr(p.drop_thr ? p.drop_thr : 1u) {}
#ifdef FASN_F32_NODROP
    FASN_DEV bool keep(int, int, int) const { return true; }
#else
    FASN_DEV bool keep(int bh, int row, int key) const { return drop_keep_at(dsd, (uint32_t)bh, (uint32_t)row, (uint32_t)key, thr); }
#endif
};

// ------------------------------------------------------------------------------------------------ forward
template <int D, int MODE>
__global__ void __launch_bounds__(256) fasn_f32_fwd_kernel(const FwdParams p) {
    constexpr int BM = 128, TILEB = 64 * D * 4, KC = D / 8, DB = D / 32;
    using St = Stage32<D>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const ldsK = smem;
    char* const ldsV = smem + 2 * TILEB;
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, hi = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int bh, qi;
    block_to_work(blockIdx.x, p.B * p.H, p.nqblk, bh, qi);
    constexpr bool GEN = MODE == MODE_GENERAL_SLOW;
    const F32Drop fdrop(p);
    const bool causal = MODE == MODE_CAUSAL || (GEN && p.causal);
    const int qblk = causal ? (p.nqblk - 1 - qi) : qi;
    const int b = bh / p.H, h = bh % p.H;
    const int q0 = qblk * BM, qw0 = q0 + wave * 32, row = qw0 + l31, coff = p.Sk - p.Sq;
    GenElem ge;
    if (GEN) ge.init(p, b, h);
    const char* qbase = p.q + (b * p.qs[0] + h * p.qs[1]) * 4;
    const char* kbase = p.k + (b * p.ks[0] + (h / p.kvg) * p.ks[1]) * 4;
    const char* vbase = p.v + (b * p.vs[0] + (h / p.kvg) * p.vs[1]) * 4;
    int ntiles = (p.Sk + 63) / 64;
    if (causal) {
        const int kmax = min(q0 + BM, p.Sq) - 1 + coff;
        ntiles = min(ntiles, kmax < 0 ? 0 : (kmax / 64 + 1));
    }
    f32x4 qf[KC];
#pragma unroll
    for (int c = 0; c < KC; ++c) {
        qf[c] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (row < p.Sq) qf[c] = *reinterpret_cast<const f32x4*>(qbase + (int64_t)row * p.qs[2] * 4 + (2 * c + hi) * 16);
    }
    const __amdgpu_buffer_rsrc_t krs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(kbase), 0, p.kbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t vrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(vbase), 0, p.vbytes, 0x00020000);
    St sK, sV;
    sK.init(tid, p.ks[2]);
    sV.init(tid, p.vs[2]);
    u32x4 stg[St::NLD];   // one staging set: the next tile's K rows travel during the first S chain, its V rows during the second
    const bool sink = p.n > 0.f;
    float m_run = sink ? 0.f : -INFINITY, l_run = (sink && hi == 0) ? p.n : 0.f;
    f32x16 oacc[DB];
#pragma unroll
    for (int d = 0; d < DB; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[d][r] = 0.f;
    if (ntiles > 0) {
        sK.gload(stg, krs, 0, p.ks[2]);
        sK.lstore(stg, ldsK);
        sV.gload(stg, vrs, 0, p.vs[2]);
        sV.lstore(stg, ldsV);
    }
    __syncthreads();
#pragma unroll
    for (int c = 0; c < KC; ++c) retire_loads(qf[c]);
    const int vis = causal ? (row + coff) : 0x7fffffff;
    for (int t = 0; t < ntiles; ++t) {
        const int buf = t & 1, k0 = t * 64;
        sK.gload(stg, krs, k0 + 64, p.ks[2]);
        const char* tK = ldsK + buf * TILEB;
        const char* tV = ldsV + buf * TILEB;
        // LDS addresses from a fresh copy of the lane id per tile: hoisted out of the loop, the swizzled addresses of a tile's reads are what spilled
        const int lane_f = fresh_lane_id(), l31 = lane_f & 31, hi = lane_f >> 5;
        f32x16 sacc[2];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
            for (int r = 0; r < 16; ++r) sacc[kb][r] = 0.f;
#pragma unroll
            for (int c = 0; c < KC; ++c) {
                const f32x4 kf = lds_row4<D>(tK, kb * 32 + l31, 2 * c + hi);
#pragma unroll
                for (int e = 0; e < 4; ++e) sacc[kb] = mfma32(kf[e], qf[c][e], sacc[kb]);
                if (D >= 128 && (c & 3) == 3) asm volatile("" ::: "memory");
            }
            // the other buffers were last read before the barrier that ended the previous tile: the staged rows go there as soon as a chain is done
            if (kb == 0) {
                sK.lstore(stg, ldsK + (buf ^ 1) * TILEB);
                sV.gload(stg, vrs, k0 + 64, p.vs[2]);
            } else {
                sV.lstore(stg, ldsV + (buf ^ 1) * TILEB);
            }
        }
        // online softmax_n (log2 domain), exact every tile: the matrix pipe is the bottleneck here, not the VALU
        float mx = -INFINITY;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = k0 + kb * 32 + acc_row(r, hi);
                bool show = key < p.Sk && key <= vis;
                float y = sacc[kb][r] * p.c;
                if (GEN && show && row < p.Sq) show = ge.apply(p, row, key, y);
                y = show ? y : -INFINITY;
                sacc[kb][r] = y;
                mx = fmaxf(mx, y);
            }
        mx = max_across_halves(mx);
        const float m_new = fmaxf(m_run, mx);
        const float m_use = (m_new == -INFINITY) ? 0.f : m_new;
        const float alpha = fast_exp2(m_run - m_use);
        float rs = 0.f;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                sacc[kb][r] = fast_exp2(sacc[kb][r] - m_use);
                rs += sacc[kb][r];   // the row sum keeps the undropped weights
            }
        if (GEN && p.drop_thr) {   // (a pass of its own behind a wave-uniform branch: with the hash inside the element loop above hipcc 7.2 miscompiled the
                                   // p = 0 path of the D = 64 instantiation - garbage in the rows of a ragged last query block, round 6)
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (!fdrop.keep(bh, row, k0 + kb * 32 + acc_row(r, hi))) sacc[kb][r] = 0.f;
        }
        l_run = l_run * alpha + rs;
        m_run = m_new;
        if (!__all(alpha == 1.0f)) {
#pragma unroll
            for (int d = 0; d < DB; ++d)
#pragma unroll
                for (int r = 0; r < 16; ++r) oacc[d][r] *= alpha;
        }
        // O^T[d][q] += V^T[d][key] P^T[key][q]: MFMA r contracts the two keys acc_row(r,0), acc_row(r,1)
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r)
#pragma unroll
                for (int d = 0; d < DB; ++d) {
                    const float vf = lds_elem32<D>(tV, kb * 32 + acc_row(r, hi), d * 32 + l31);
                    oacc[d] = mfma32(vf, sacc[kb][r], oacc[d]);
                    if (D >= 128 && d == DB - 1 && (r & 3) == 3) asm volatile("" ::: "memory");
                }
        __syncthreads();
    }
    const float l_tot = sum_across_halves(l_run);
    const float inv = l_tot > 0.f ? ((GEN && p.drop_thr) ? p.drop_scale : 1.0f) / l_tot : 0.f;
    if (row < p.Sq) {
        if (p.lse != nullptr && hi == 0) {
            const float m_use = (m_run == -INFINITY) ? 0.f : m_run;
            p.lse[(int64_t)bh * p.Sq + row] = l_tot > 0.f ? (m_use + __builtin_log2f(l_tot)) * kLn2 : -INFINITY;
        }
        char* rp = p.o + (b * p.os[0] + h * p.os[1] + (int64_t)row * p.os[2]) * 4;
#pragma unroll
        for (int d = 0; d < DB; ++d)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4 x;
#pragma unroll
                for (int e = 0; e < 4; ++e) x[e] = oacc[d][4 * g + e] * inv;
                *reinterpret_cast<f32x4*>(rp + (d * 32 + 8 * g + 4 * hi) * 4) = x;
            }
    }
}

// ------------------------------------------------------------------------------------------------ delta = rowsum(O o dO)
template <int D>
__global__ void __launch_bounds__(256) fasn_f32_delta_kernel(const BwdParams p) {
    constexpr int LPR = D / 4, RPB = 256 / LPR;
    const int tid = threadIdx.x, sub = tid % LPR;
    const int64_t rows = (int64_t)p.f.B * p.f.H * p.f.Sq;
    const int64_t gr = (int64_t)blockIdx.x * RPB + tid / LPR;
    float acc = 0.f;
    if (gr < rows) {
        const int i = (int)(gr % p.f.Sq), bh = (int)(gr / p.f.Sq), b = bh / p.f.H, h = bh % p.f.H;
        const f32x4 a = *reinterpret_cast<const f32x4*>(p.f.o + (b * p.f.os[0] + h * p.f.os[1] + (int64_t)i * p.f.os[2]) * 4 + sub * 16);
        const f32x4 d = *reinterpret_cast<const f32x4*>(p.dout + (b * p.dos[0] + h * p.dos[1] + (int64_t)i * p.dos[2]) * 4 + sub * 16);
        acc = a[0] * d[0] + a[1] * d[1] + a[2] * d[2] + a[3] * d[3];
    }
#pragma unroll
    for (int o = 1; o < LPR; o <<= 1) acc += __shfl_xor(acc, o);
    if (gr < rows && sub == 0) p.delta[gr] = acc;
}

// ------------------------------------------------------------------------------------------------ dQ
template <int D, int MODE>
__global__ void __launch_bounds__(256) fasn_f32_dq_kernel(const BwdParams bp) {
    const FwdParams& p = bp.f;
    constexpr int BM = 128, TILEB = 64 * D * 4, KC = D / 8, DB = D / 32;
    using St = Stage32<D>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const ldsK = smem;
    char* const ldsV = smem + 2 * TILEB;
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, hi = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int bh, qi;
    block_to_work(blockIdx.x, p.B * p.H, bp.nblk, bh, qi);
    constexpr bool GEN = MODE == MODE_GENERAL_SLOW;
    const F32Drop fdrop(bp.f);
    const bool causal = MODE == MODE_CAUSAL || (GEN && p.causal);
    const int qblk = causal ? (bp.nblk - 1 - qi) : qi;
    const int b = bh / p.H, h = bh % p.H;
    const int q0 = qblk * BM, qw0 = q0 + wave * 32, row = qw0 + l31, coff = p.Sk - p.Sq;
    GenElem ge;
    if (GEN) ge.init(p, b, h);
    const char* kbase = p.k + (b * p.ks[0] + (h / p.kvg) * p.ks[1]) * 4;
    const char* vbase = p.v + (b * p.vs[0] + (h / p.kvg) * p.vs[1]) * 4;
    int ntiles = (p.Sk + 63) / 64;
    if (causal) {
        const int kmax = min(q0 + BM, p.Sq) - 1 + coff;
        ntiles = min(ntiles, kmax < 0 ? 0 : (kmax / 64 + 1));
    }
    f32x4 qf[KC], dof[KC];
    const bool ok = row < p.Sq;
#pragma unroll
    for (int c = 0; c < KC; ++c) {
        qf[c] = dof[c] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (ok) {
            qf[c] = *reinterpret_cast<const f32x4*>(p.q + (b * p.qs[0] + h * p.qs[1] + (int64_t)row * p.qs[2]) * 4 + (2 * c + hi) * 16);
            dof[c] = *reinterpret_cast<const f32x4*>(bp.dout + (b * bp.dos[0] + h * bp.dos[1] + (int64_t)row * bp.dos[2]) * 4 + (2 * c + hi) * 16);
        }
    }
    float lse2 = ok ? p.lse[(int64_t)bh * p.Sq + row] : 0.f;
    lse2 = (lse2 == -INFINITY) ? INFINITY : lse2 * kLog2e;
    float dlt = ok ? bp.delta[(int64_t)bh * p.Sq + row] : 0.f;
    const __amdgpu_buffer_rsrc_t krs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(kbase), 0, p.kbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t vrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(vbase), 0, p.vbytes, 0x00020000);
    St sK, sV;
    sK.init(tid, p.ks[2]);
    sV.init(tid, p.vs[2]);
    u32x4 stg[St::NLD];   // one staging set: the next tile's K rows travel during the first S / dP chain, its V rows during the second
    f32x16 dqacc[DB];
#pragma unroll
    for (int d = 0; d < DB; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) dqacc[d][r] = 0.f;
    if (ntiles > 0) {
        sK.gload(stg, krs, 0, p.ks[2]);
        sK.lstore(stg, ldsK);
        sV.gload(stg, vrs, 0, p.vs[2]);
        sV.lstore(stg, ldsV);
    }
    __syncthreads();
#pragma unroll
    for (int c = 0; c < KC; ++c) {
        retire_loads(qf[c]);
        retire_loads(dof[c]);
    }
    retire_loads(lse2);
    retire_loads(dlt);
    const int vis = causal ? (row + coff) : 0x7fffffff;
    for (int t = 0; t < ntiles; ++t) {
        const int buf = t & 1, k0 = t * 64;
        sK.gload(stg, krs, k0 + 64, p.ks[2]);
        const char* tK = ldsK + buf * TILEB;
        const char* tV = ldsV + buf * TILEB;
        // LDS addresses from a fresh copy of the lane id per tile: hoisted out of the loop, the swizzled addresses of a tile's reads are what spilled
        const int lane_f = fresh_lane_id(), l31 = lane_f & 31, hi = lane_f >> 5;
        f32x16 sacc[2], pacc[2];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
            for (int r = 0; r < 16; ++r) sacc[kb][r] = pacc[kb][r] = 0.f;
#pragma unroll
            for (int c = 0; c < KC; ++c) {
                const f32x4 kf = lds_row4<D>(tK, kb * 32 + l31, 2 * c + hi);
                const f32x4 vf = lds_row4<D>(tV, kb * 32 + l31, 2 * c + hi);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    sacc[kb] = mfma32(kf[e], qf[c][e], sacc[kb]);
                    pacc[kb] = mfma32(vf[e], dof[c][e], pacc[kb]);
                }
                if (D >= 128 && (c & 3) == 3) asm volatile("" ::: "memory");
            }
            // the other buffers were last read before the barrier that ended the previous tile: the staged rows go there as soon as a chain is done
            if (kb == 0) {
                sK.lstore(stg, ldsK + (buf ^ 1) * TILEB);
                sV.gload(stg, vrs, k0 + 64, p.vs[2]);
            } else {
                sV.lstore(stg, ldsV + (buf ^ 1) * TILEB);
            }
        }
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = k0 + kb * 32 + acc_row(r, hi);
                bool show = key < p.Sk && key <= vis;
                float y = sacc[kb][r] * p.c;
                if (GEN && show && ok) show = ge.apply(p, row, key, y);
                float pv = fast_exp2(y - lse2);
                pv = show ? pv : 0.f;
                float dp = pacc[kb][r];
                if (GEN && p.drop_thr) dp = fdrop.keep(bh, row, key) ? dp * p.drop_scale : 0.f;
                sacc[kb][r] = pv * (dp - dlt);   // dS^T (without the scale factor)
                if (GEN && bp.dbias != nullptr && ok && key < p.Sk)
                    reinterpret_cast<float*>(bp.dbias)[b * bp.dbs[0] + h * bp.dbs[1] + (int64_t)row * bp.dbs[2] + key] = sacc[kb][r];
            }
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r)
#pragma unroll
                for (int d = 0; d < DB; ++d) {
                    const float kt = lds_elem32<D>(tK, kb * 32 + acc_row(r, hi), d * 32 + l31);
                    dqacc[d] = mfma32(kt, sacc[kb][r], dqacc[d]);
                    if (D >= 128 && d == DB - 1 && (r & 3) == 3) asm volatile("" ::: "memory");
                }
        __syncthreads();
    }
    if (ok) {
        char* rp = bp.dq + (b * bp.dqs[0] + h * bp.dqs[1] + (int64_t)row * bp.dqs[2]) * 4;
#pragma unroll
        for (int d = 0; d < DB; ++d)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4 x;
#pragma unroll
                for (int e = 0; e < 4; ++e) x[e] = dqacc[d][4 * g + e] * bp.scale;
                *reinterpret_cast<f32x4*>(rp + (d * 32 + 8 * g + 4 * hi) * 4) = x;
            }
    }
}

// ------------------------------------------------------------------------------------------------ dK, dV
template <int D, int MODE>
__global__ void __launch_bounds__(256) fasn_f32_dkdv_kernel(const BwdParams bp) {
    const FwdParams& p = bp.f;
    constexpr int BN = 128, TILEB = 64 * D * 4, KC = D / 8, DB = D / 32;
    using St = Stage32<D>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const ldsQ = smem;
    char* const ldsDO = smem + 2 * TILEB;
    float* const ldsLse = reinterpret_cast<float*>(smem + 4 * TILEB);
    float* const ldsDlt = ldsLse + 2 * 64;
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, hi = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // grouped-query attention: one workgroup per (batch, K/V head, key block) walks all query heads of the group (dK / dV per K/V head)
    const int kvg = p.kvg, Hkv = p.H / kvg;
    int bhk, kblk;
    block_to_work(blockIdx.x, p.B * Hkv, bp.nblk, bhk, kblk);
    constexpr bool GEN = MODE == MODE_GENERAL_SLOW;
    const F32Drop fdrop(bp.f);
    const bool causal = MODE == MODE_CAUSAL || (GEN && p.causal);
    const int b = bhk / Hkv, hk = bhk % Hkv;
    const int kw0 = kblk * BN + wave * 32, key = kw0 + l31, coff = p.Sk - p.Sq;
    f32x16 dkacc[DB], dvacc[DB];
#pragma unroll
    for (int d = 0; d < DB; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) dkacc[d][r] = dvacc[d][r] = 0.f;
    for (int g = 0; g < kvg; ++g) {
    const int h = hk * kvg + g, bh = b * p.H + h;
    if (g > 0) __syncthreads();
    GenElem ge;
    if (GEN) ge.init(p, b, h);
    const char* qbase = p.q + (b * p.qs[0] + h * p.qs[1]) * 4;
    const char* dobase = bp.dout + (b * bp.dos[0] + h * bp.dos[1]) * 4;
    const float* lsebase = p.lse + (int64_t)bh * p.Sq;
    const float* dltbase = bp.delta + (int64_t)bh * p.Sq;
    const int ntq = (p.Sq + 63) / 64;
    int tq0 = 0;
    if (causal) {
        const int first_row = kblk * BN - coff;
        tq0 = first_row <= 0 ? 0 : first_row / 64;
    }
    f32x4 kf[KC], vf[KC];
    const bool ok = key < p.Sk;
    {   // the lane's K / V row as range-checked buffer loads (rows behind the last key read as zero; `ok` covers a length-1 sequence with row stride 0)
        const __amdgpu_buffer_rsrc_t krs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(p.k + (b * p.ks[0] + hk * p.ks[1]) * 4), 0, p.kbytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t vrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(p.v + (b * p.vs[0] + hk * p.vs[1]) * 4), 0, p.vbytes, 0x00020000);
        const unsigned ko = (unsigned)(key * (int)p.ks[2] * 4 + hi * 16), vo = (unsigned)(key * (int)p.vs[2] * 4 + hi * 16);
#pragma unroll
        for (int c = 0; c < KC; ++c) {
            kf[c] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(krs, ko + c * 32, 0, 0));
            vf[c] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(vrs, vo + c * 32, 0, 0));
            if (!ok) kf[c] = vf[c] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
    }
    const __amdgpu_buffer_rsrc_t qrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(qbase), 0, bp.qbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t drs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(dobase), 0, bp.dobytes, 0x00020000);
    St sQ, sD;
    sQ.init(tid, p.qs[2]);
    sD.init(tid, bp.dos[2]);
    u32x4 stg[St::NLD];   // one staging set: a tile's Q rows travel during the first S / dP chain, its dO rows during the second
    float stL = 0.f, stX = 0.f;
    auto stats_gload = [&](int row0) {
        if (tid < 64) {
            const int gr = row0 + tid;
            float l = 0.f, x = 0.f;
            if (gr < p.Sq) {
                l = lsebase[gr];
                x = dltbase[gr];
            }
            stL = (l == -INFINITY) ? INFINITY : l * kLog2e;
            stX = x;
        }
    };
    auto stats_lstore = [&](int buf) {
        if (tid < 64) {
            ldsLse[buf * 64 + tid] = stL;
            ldsDlt[buf * 64 + tid] = stX;
        }
    };
    if (tq0 < ntq) {
        sQ.gload(stg, qrs, tq0 * 64, p.qs[2]);
        stats_gload(tq0 * 64);
        sQ.lstore(stg, ldsQ);
        sD.gload(stg, drs, tq0 * 64, bp.dos[2]);
        sD.lstore(stg, ldsDO);
        stats_lstore(0);
    }
    __syncthreads();
#pragma unroll
    for (int c = 0; c < KC; ++c) {
        retire_loads(kf[c]);
        retire_loads(vf[c]);
    }
    for (int tq = tq0; tq < ntq; ++tq) {
        const int buf = (tq - tq0) & 1, r0 = tq * 64;
        sQ.gload(stg, qrs, r0 + 64, p.qs[2]);
        stats_gload(r0 + 64);
        const char* tQ = ldsQ + buf * TILEB;
        const char* tD = ldsDO + buf * TILEB;
        const float* tL = ldsLse + buf * 64;
        const float* tX = ldsDlt + buf * 64;
        // LDS addresses from a fresh copy of the lane id per tile: hoisted out of the loop, the 16 + 16 swizzled addresses of a tile's reads are what spilled
        const int lane_f = fresh_lane_id(), l31 = lane_f & 31, hi = lane_f >> 5;
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) {
            f32x16 sacc, pacc;
#pragma unroll
            for (int r = 0; r < 16; ++r) sacc[r] = pacc[r] = 0.f;
#pragma unroll
            for (int c = 0; c < KC; ++c) {
                const f32x4 qa = lds_row4<D>(tQ, qb * 32 + l31, 2 * c + hi);
                const f32x4 da = lds_row4<D>(tD, qb * 32 + l31, 2 * c + hi);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    sacc = mfma32(qa[e], kf[c][e], sacc);   // S[q][key]
                    pacc = mfma32(da[e], vf[c][e], pacc);   // dP[q][key]
                }
                if (D >= 128 && (c & 3) == 3) asm volatile("" ::: "memory");
            }
            // the other buffer was last read before the barrier that ended the previous tile: the staged rows go there as soon as the chain is done
            if (qb == 0) {
                sQ.lstore(stg, ldsQ + (buf ^ 1) * TILEB);
                sD.gload(stg, drs, r0 + 64, bp.dos[2]);
            } else {
                sD.lstore(stg, ldsDO + (buf ^ 1) * TILEB);
                stats_lstore(buf ^ 1);
            }
            // dropout in passes of their own behind a wave-uniform branch (see the forward): dP is dropped and scaled in front of the element
            // pass, the weights that feed dV behind it (dS uses the undropped P); the keep bits are computed twice - the fp32 dropout path is
            // the reference's test grid, not a hot path
            if (GEN && p.drop_thr) {
                int r0a = r0 + qb * 32;   // (opaque copies here and below: otherwise the hash inputs of all four passes of a tile are computed at its top and parked in scratch)
                asm volatile("" : "+s"(r0a));
#pragma unroll
                for (int r = 0; r < 16; ++r) pacc[r] *= fdrop.keep(bh, r0a + acc_row(r, hi), key) ? p.drop_scale : 0.f;
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int rr = qb * 32 + acc_row(r, hi);
                const int row = r0 + rr;
                bool show = ok && row < p.Sq && (!causal || key <= row + coff);
                float y = sacc[r] * p.c;
                if (GEN && show) show = ge.apply(p, row, key, y);
                float pv = fast_exp2(y - tL[rr]);
                pv = show ? pv : 0.f;
                sacc[r] = pv;
                pacc[r] = pv * (pacc[r] - tX[rr]);      // dS uses the undropped P
                if (GEN && D >= 128 && (r & 3) == 3) asm volatile("" ::: "memory");   // (keeps the per-element mask / bias loads of four rows together instead of all sixteen in flight)
            }
            if (GEN && p.drop_thr) {
                int r0b = r0 + qb * 32;
                asm volatile("" : "+s"(r0b));
#pragma unroll
                for (int r = 0; r < 16; ++r) sacc[r] *= fdrop.keep(bh, r0b + acc_row(r, hi), key) ? p.drop_scale : 0.f;   // dropped weights feed dV
            }
#pragma unroll
            for (int r = 0; r < 16; ++r)
#pragma unroll
                for (int d = 0; d < DB; ++d) {
                    const int rr = qb * 32 + acc_row(r, hi);
                    const float dot = lds_elem32<D>(tD, rr, d * 32 + l31);
                    const float qt = lds_elem32<D>(tQ, rr, d * 32 + l31);
                    dvacc[d] = mfma32(dot, sacc[r], dvacc[d]);
                    dkacc[d] = mfma32(qt, pacc[r], dkacc[d]);
                    if (D >= 128 && d == DB - 1 && (r & 3) == 3) asm volatile("" ::: "memory");
                }
        }
        __syncthreads();
    }
    }   // query heads of the group
    if (key < p.Sk) {
        char* rk = bp.dk + (b * bp.dks[0] + hk * bp.dks[1] + (int64_t)key * bp.dks[2]) * 4;   // dK / dV are [B, H / kvg, Sk, D]
        char* rv = bp.dv + (b * bp.dvs[0] + hk * bp.dvs[1] + (int64_t)key * bp.dvs[2]) * 4;
#pragma unroll
        for (int d = 0; d < DB; ++d)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4 x, y;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    x[e] = dkacc[d][4 * g + e] * bp.scale;
                    y[e] = dvacc[d][4 * g + e];
                }
                *reinterpret_cast<f32x4*>(rk + (d * 32 + 8 * g + 4 * hi) * 4) = x;
                *reinterpret_cast<f32x4*>(rv + (d * 32 + 8 * g + 4 * hi) * 4) = y;
            }
    }
}

}  // namespace fasn

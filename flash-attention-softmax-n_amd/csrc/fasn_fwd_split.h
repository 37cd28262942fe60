// fasn_fwd_split.h — forward (MODE_PLAIN / MODE_CAUSAL) with the 64-key tile processed as two 32-key sub-tiles so that a
// wave's own MFMAs run under its own exponentials:
//
//     QK(kb0) | QK(kb1) issued, then softmax(kb0) | PV(kb0) issued, then softmax(kb1) | PV(kb1)
//
// The optimistic softmax (exp2 against the current running max, no tile max) has no reduction before the exponentials, so
// sub-tile kb0 can be exponentiated while the QK^T MFMAs of kb1 are still in the matrix pipe, and P(kb0) V(kb0) runs
// while kb1 is exponentiated (tools/ubench3: one wave alternating 16 independent MFMAs with this VALU mix needs 414 ns per
// round instead of the 625 ns of the two phases run back to back). The overflow guard is checked per sub-tile BEFORE its
// PV MFMAs are issued; a failing or partly hidden sub-tile takes the exact path (max, re-centre, rescale) for that
// sub-tile only - the accumulator then already contains every earlier sub-tile at the old max, which is what the rescale
// assumes. Math, layouts, staging (two-set ring) and epilogue are those of fasn_fwd_kernel.h.
#pragma once
#include "fasn_fwd_kernel.h"

namespace fasn {

template <typename Tag, int D, int QB, int MODE, int OCC>
__global__ void __launch_bounds__(256, OCC) fasn_fwd_split_kernel(const FwdParams p) {
    static_assert(MODE == MODE_PLAIN || MODE == MODE_CAUSAL, "masked / biased attention uses fasn_fwd_kernel");
    using E = ET<Tag>;
    using vec8 = typename E::vec8;
    constexpr int NT = 256;
    constexpr int BM = 4 * QB * 32;
    constexpr int ROWB = D * 2;
    constexpr int TILEB = KT * ROWB;
    constexpr int KS = D / 16;
    constexpr int DB = D / 32;
    constexpr int CPR = D / 8;
    constexpr int NLD = (KT * CPR) / NT;

    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const ldsK = smem;
    char* const ldsV = smem + 2 * TILEB;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31;
    const int hi = lane >> 5;

    int bh, qi;
    block_to_work(blockIdx.x, p.B * p.H, p.nqblk, bh, qi);
    constexpr bool causal = MODE == MODE_CAUSAL;
    const int qblk = causal ? (p.nqblk - 1 - qi) : qi;
    const int b = bh / p.H, h = bh % p.H;
    const int q0 = qblk * BM;
    const int qw0 = q0 + wave * (QB * 32);
    const char* qbase = p.q + (b * p.qs[0] + h * p.qs[1]) * 2;
    const char* kbase = p.k + (b * p.ks[0] + (h / p.kvg) * p.ks[1]) * 2;
    const char* vbase = p.v + (b * p.vs[0] + (h / p.kvg) * p.vs[1]) * 2;
    const int coff = p.Sk - p.Sq;

    int ntiles = (p.Sk + KT - 1) / KT;
    if (causal) {
        const int kmax = min(q0 + BM, p.Sq) - 1 + coff;
        ntiles = min(ntiles, kmax < 0 ? 0 : (kmax / KT + 1));
    }

    vec8 qf[QB][KS];
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
        const int row = qw0 + qb * 32 + l31;
        const bool ok = row < p.Sq;
        const char* rp = qbase + (int64_t)row * p.qs[2] * 2 + hi * 16;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            u32x4 raw = {0u, 0u, 0u, 0u};
            if (ok) raw = gload16(rp + s * 32);
            __builtin_memcpy(&qf[qb][s], &raw, 16);
        }
    }

    const __amdgpu_buffer_rsrc_t krs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(kbase), 0, p.kbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t vrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(vbase), 0, p.vbytes, 0x00020000);
    unsigned kvoff[NLD], vvoff[NLD];
    int ldsoff[NLD];
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
        const int ci = tid + i * NT;
        const int row = ci / CPR, ch = ci % CPR;
        kvoff[i] = (unsigned)(row * (int)p.ks[2] * 2 + ch * 16);
        vvoff[i] = (unsigned)(row * (int)p.vs[2] * 2 + ch * 16);
        ldsoff[i] = tile_off<D>(row, ch);
    }
    const int ktile_bytes = KT * (int)p.ks[2] * 2;
    const int vtile_bytes = KT * (int)p.vs[2] * 2;
    u32x4 stK[2][NLD], stV[2][NLD];
    auto stage_load = [&](int t, auto SET) {
        constexpr int S_ = decltype(SET)::value;
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            stK[S_][i] = __builtin_amdgcn_raw_buffer_load_b128(krs, kvoff[i], t * ktile_bytes, 0);
            stV[S_][i] = __builtin_amdgcn_raw_buffer_load_b128(vrs, vvoff[i], t * vtile_bytes, 0);
        }
    };
    auto stage_store = [&](int buf, auto SET) {
        constexpr int S_ = decltype(SET)::value;
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            *LDS_PTR(u32x4, ldsK + buf * TILEB + ldsoff[i]) = stK[S_][i];
            *LDS_PTR(u32x4, ldsV + buf * TILEB + ldsoff[i]) = stV[S_][i];
        }
    };
    using Set0 = std::integral_constant<int, 0>;
    using Set1 = std::integral_constant<int, 1>;

    float m_run[QB], l_run[QB];
    f32x16 oacc[QB][DB];
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
        const bool sink = p.n > 0.f;
        m_run[qb] = sink ? 0.f : -INFINITY;
        l_run[qb] = (sink && hi == 0) ? p.n : 0.f;
#pragma unroll
        for (int d = 0; d < DB; ++d)
#pragma unroll
            for (int r = 0; r < 16; ++r) oacc[qb][d][r] = 0.f;
    }

    if (ntiles > 0) {
        stage_load(0, Set0{});
        stage_store(0, Set0{});
        stage_load(1, Set1{});
    }
    __syncthreads();
#pragma unroll
    for (int qb = 0; qb < QB; ++qb)
#pragma unroll
        for (int s = 0; s < KS; ++s) retire_loads(qf[qb][s]);

    const int wave_first_vis = qw0 + coff;
    const int wave_last_vis = qw0 + QB * 32 - 1 + coff;

    auto tile_body = [&](const int t, auto LSET, auto SSET) {
        const int buf = t & 1;
        const int k0 = t * KT;
        stage_load(t + 2, LSET);   // past-the-end tiles read back as zeros
        const bool skip = causal && (k0 > wave_last_vis);
        if (!skip) {
            const char* tK = ldsK + buf * TILEB;
            const char* tV = ldsV + buf * TILEB;
            f32x16 sacc[QB][2];
            vec8 pf[QB][2];  // [qb][t2] of the sub-tile being processed
            auto qk = [&](int kb) {
#pragma unroll
                for (int qb = 0; qb < QB; ++qb)
#pragma unroll
                    for (int r = 0; r < 16; ++r) sacc[qb][kb][r] = 0.f;
#pragma unroll
                for (int s = 0; s < KS; ++s) {
                    vec8 kf = lds_read_rowfrag<E, D>(tK, kb * 32 + l31, s, hi);
#pragma unroll
                    for (int qb = 0; qb < QB; ++qb) sacc[qb][kb] = E::mfma(kf, qf[qb][s], sacc[qb][kb]);
                }
            };
            qk(0);
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
                const int kk0 = k0 + kb * 32;
                bool exact = (kk0 + 32 > p.Sk);
                if (causal) exact = exact || ((kk0 + 31) > wave_first_vis);
                if (kb == 0) qk(1);   // in the matrix pipe while sub-tile 0 is exponentiated
                float rs[QB];
                bool bad = false;
                if (!exact) {
#pragma unroll
                    for (int qb = 0; qb < QB; ++qb) {
                        float sum = 0.f;
                        const float mneg = -m_run[qb];
#pragma unroll
                        for (int t2 = 0; t2 < 2; ++t2) {
                            f32x8 x;
#pragma unroll
                            for (int e = 0; e < 8; ++e) {
                                x[e] = fast_exp2(__builtin_fmaf(sacc[qb][kb][8 * t2 + e], p.c, mneg));
                                sum += x[e];
                            }
                            pf[qb][t2] = E::cvt8(x);
                        }
                        rs[qb] = sum;
                        bad = bad || !(sum <= kSumLimit * 0.5f);   // 16 values per lane and sub-tile
                    }
                    if (__any(bad)) exact = true;
                }
                if (exact) {
#pragma unroll
                    for (int qb = 0; qb < QB; ++qb) {
                        const int row = qw0 + qb * 32 + l31;
                        const int vis = causal ? (row + coff) : 0x7fffffff;
                        float mx = -INFINITY;
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int key = kk0 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                            const bool show = (key < p.Sk) && (key <= vis);
                            const float y = show ? sacc[qb][kb][r] * p.c : -INFINITY;
                            sacc[qb][kb][r] = y;
                            mx = fmaxf(mx, y);
                        }
                        mx = max_across_halves(mx);
                        const float m_new = fmaxf(m_run[qb], mx);
                        const float m_use = (m_new == -INFINITY) ? 0.f : m_new;
                        const float alpha = fast_exp2(m_run[qb] - m_use);
                        float sum = 0.f;
#pragma unroll
                        for (int t2 = 0; t2 < 2; ++t2) {
                            f32x8 x;
#pragma unroll
                            for (int e = 0; e < 8; ++e) {
                                x[e] = fast_exp2(sacc[qb][kb][8 * t2 + e] - m_use);
                                sum += x[e];
                            }
                            pf[qb][t2] = E::cvt8(x);
                        }
                        l_run[qb] = l_run[qb] * alpha + sum;
                        m_run[qb] = m_new;
                        if (!__all(alpha == 1.0f)) {
#pragma unroll
                            for (int d = 0; d < DB; ++d)
#pragma unroll
                                for (int r = 0; r < 16; ++r) oacc[qb][d][r] *= alpha;
                        }
                    }
                } else {
#pragma unroll
                    for (int qb = 0; qb < QB; ++qb) l_run[qb] += rs[qb];
                }
                // O^T += V(kb)^T P(kb)^T
#pragma unroll
                for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
                    for (int d = 0; d < DB; ++d) {
                        vec8 vf = lds_read_trfrag<E, D>(tV, kb * 32 + 16 * t2, d, lane);
#pragma unroll
                        for (int qb = 0; qb < QB; ++qb) oacc[qb][d] = E::mfma(vf, pf[qb][t2], oacc[qb][d]);
                    }
            }
        }
        stage_store(buf ^ 1, SSET);
        __syncthreads();
    };
    for (int t = 0; t < ntiles; t += 2) {
        tile_body(t, Set0{}, Set1{});
        if (t + 1 < ntiles) tile_body(t + 1, Set1{}, Set0{});
    }

    char* obase = p.o + (b * p.os[0] + h * p.os[1]) * 2;
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
        const int row = qw0 + qb * 32 + l31;
        const float l_tot = sum_across_halves(l_run[qb]);
        const float inv = l_tot > 0.f ? 1.0f / l_tot : 0.f;
        if (row < p.Sq) {
            if (p.lse != nullptr && hi == 0) {
                const float m_use = (m_run[qb] == -INFINITY) ? 0.f : m_run[qb];
                p.lse[(int64_t)bh * p.Sq + row] = l_tot > 0.f ? (m_use + __builtin_log2f(l_tot)) * kLn2 : -INFINITY;
            }
            char* rp = obase + (int64_t)row * p.os[2] * 2;
#pragma unroll
            for (int d = 0; d < DB; ++d)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    f32x4 x;
#pragma unroll
                    for (int e = 0; e < 4; ++e) x[e] = oacc[qb][d][4 * g + e] * inv;
                    typename E::vec4 y = E::cvt4(x);
                    u32x2 raw;
                    __builtin_memcpy(&raw, &y, 8);
                    gstore8(rp + (d * 32 + 8 * g + 4 * hi) * 2, raw);
                }
        }
    }
}

}  // namespace fasn

// fasn_fwd_pp.h — "ping-pong" forward (MODE_PLAIN / MODE_CAUSAL): 8 waves per workgroup in two groups of four.
//
// Measured on MI355X (tools/ubench2.cpp): an MFMA-only wave and a VALU/transcendental-only wave on the SAME SIMD overlap
// perfectly, but two waves that run the same QK^T -> softmax -> PV sequence fall into lock step (both in the matrix
// phase, then both in the exp phase) and their times add. Here the phases are forced apart: waves w and w+4 share a
// SIMD (a workgroup's waves are dealt to the 4 SIMDs cyclically), group A = waves 0-3, group B = waves 4-7, and every
// phase boundary is an s_barrier of the whole workgroup:
//
//     phase:   0        1        2        3        4
//     A:     QK(0) | SM(0)  | M(0)   | SM(1)  | M(1)   ...        SM(t) = softmax_n of tile t (VALU / v_exp)
//     B:      -    | QK(0)  | SM(0)  | M(0)   | SM(1)  ...        M(t)  = PV(t) + QK(t+1)      (MFMA + LDS reads)
//
// so on each SIMD one wave is always in a matrix phase while its partner is in an exponential phase.
// LDS holds two "pairs" {K(t+1), V(t)} (what M(t) reads); pair t+1 is written by every wave during its own M(t) phase
// (A in phase 2t+2, B in 2t+3: after the last read of pair t-1, before the first read of pair t+1), from registers
// filled by buffer loads issued one iteration earlier.
// Each wave owns 32 query rows (256 rows per workgroup); math, layouts and the optimistic softmax are those of
// fasn_fwd_kernel.h.
#pragma once
#include "fasn_fwd_kernel.h"

namespace fasn {

template <typename Tag, int D, int MODE, int OCC>
__global__ void __launch_bounds__(512, OCC) fasn_fwd_pp_kernel(const FwdParams p) {
    static_assert(MODE == MODE_PLAIN || MODE == MODE_CAUSAL, "masked / biased attention uses fasn_fwd_kernel");
    static_assert(D == 64 || D == 128, "ping-pong kernel: D in {64, 128}");
    using E = ET<Tag>;
    using vec8 = typename E::vec8;
    constexpr int NT = 512;
    constexpr int BM = 8 * 32;
    constexpr int ROWB = D * 2;
    constexpr int TILEB = KT * ROWB;
    constexpr int KS = D / 16;
    constexpr int DB = D / 32;
    constexpr int CPR = D / 8;
    constexpr int NLD = (KT * CPR) / NT;

    extern __shared__ __attribute__((aligned(16))) char smem[];
    // pair buffer b: K part at smem + b*2*TILEB, V part at + TILEB
    auto bufK = [&](int b) { return smem + b * 2 * TILEB; };
    auto bufV = [&](int b) { return smem + b * 2 * TILEB + TILEB; };

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool groupB = wave >= 4;
    const int l31 = lane & 31;
    const int hi = lane >> 5;

    int bh, qi;
    block_to_work(blockIdx.x, p.B * p.H, p.nqblk, bh, qi);
    constexpr bool causal = MODE == MODE_CAUSAL;
    const int qblk = causal ? (p.nqblk - 1 - qi) : qi;
    const int b = bh / p.H, h = bh % p.H;
    const int q0 = qblk * BM;
    const int qw0 = q0 + wave * 32;

    const char* qbase = p.q + (b * p.qs[0] + h * p.qs[1]) * 2;
    const char* kbase = p.k + (b * p.ks[0] + (h / p.kvg) * p.ks[1]) * 2;
    const char* vbase = p.v + (b * p.vs[0] + (h / p.kvg) * p.vs[1]) * 2;
    const int coff = p.Sk - p.Sq;

    int ntiles = (p.Sk + KT - 1) / KT;
    if (causal) {
        const int kmax = min(q0 + BM, p.Sq) - 1 + coff;
        ntiles = min(ntiles, kmax < 0 ? 0 : (kmax / KT + 1));
    }

    vec8 qf[KS];
    {
        const int row = qw0 + l31;
        const bool ok = row < p.Sq;
        const char* rp = qbase + (int64_t)row * p.qs[2] * 2 + hi * 16;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            u32x4 raw = {0u, 0u, 0u, 0u};
            if (ok) raw = gload16(rp + s * 32);
            __builtin_memcpy(&qf[s], &raw, 16);
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
    u32x4 stK[NLD], stV[NLD];
    auto loadK = [&](int t) {
#pragma unroll
        for (int i = 0; i < NLD; ++i) stK[i] = __builtin_amdgcn_raw_buffer_load_b128(krs, kvoff[i], t * ktile_bytes, 0);
    };
    auto loadV = [&](int t) {
#pragma unroll
        for (int i = 0; i < NLD; ++i) stV[i] = __builtin_amdgcn_raw_buffer_load_b128(vrs, vvoff[i], t * vtile_bytes, 0);
    };
    auto storeK = [&](char* dst) {
#pragma unroll
        for (int i = 0; i < NLD; ++i) *LDS_PTR(u32x4, dst + ldsoff[i]) = stK[i];
    };
    auto storeV = [&](char* dst) {
#pragma unroll
        for (int i = 0; i < NLD; ++i) *LDS_PTR(u32x4, dst + ldsoff[i]) = stV[i];
    };

    const bool sink = p.n > 0.f;
    float m_run = sink ? 0.f : -INFINITY;
    float l_run = (sink && hi == 0) ? p.n : 0.f;
    f32x16 oacc[DB];
    f32x16 sacc[2];
    vec8 pf[2][2];
#pragma unroll
    for (int d = 0; d < DB; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[d][r] = 0.f;

    auto qk_tile = [&](const char* tK) {
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) sacc[kb][r] = 0.f;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                vec8 kf = lds_read_rowfrag<E, D>(tK, kb * 32 + l31, ks, hi);
                sacc[kb] = E::mfma(kf, qf[ks], sacc[kb]);
            }
    };
    auto pv_tile = [&](const char* tV) {
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
                for (int d = 0; d < DB; ++d) {
                    vec8 vf = lds_read_trfrag<E, D>(tV, kb * 32 + 16 * t2, d, lane);
                    oacc[d] = E::mfma(vf, pf[kb][t2], oacc[d]);
                }
    };

    const int wave_first_vis = qw0 + coff;
    const int row = qw0 + l31;

    // softmax_n of tile t on sacc -> pf, l_run, m_run (optimistic fast path + exact fallback)
    auto softmax_tile = [&](int t) {
        const int k0 = t * KT;
        bool exact = (k0 + KT > p.Sk);
        if (causal) exact = exact || ((k0 + KT - 1) > wave_first_vis);
        if (!exact) {
            float rs = 0.f;
            const float mneg = -m_run;
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int t2 = 0; t2 < 2; ++t2) {
                    f32x8 x;
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        x[e] = fast_exp2(__builtin_fmaf(sacc[kb][8 * t2 + e], p.c, mneg));
                        rs += x[e];
                    }
                    pf[kb][t2] = E::cvt8(x);
                }
            if (__any(!(rs <= kSumLimit))) exact = true;
            else l_run += rs;
        }
        if (exact) {
            const int vis = causal ? (row + coff) : 0x7fffffff;
            float mx = -INFINITY;
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = k0 + kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                    float y = sacc[kb][r] * p.c;
                    const bool show = (key < p.Sk) && (key <= vis);
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
                for (int t2 = 0; t2 < 2; ++t2) {
                    f32x8 x;
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        x[e] = fast_exp2(sacc[kb][8 * t2 + e] - m_use);
                        rs += x[e];
                    }
                    pf[kb][t2] = E::cvt8(x);
                }
            l_run = l_run * alpha + rs;
            m_run = m_new;
            if (!__all(alpha == 1.0f)) {
#pragma unroll
                for (int d = 0; d < DB; ++d)
#pragma unroll
                    for (int r = 0; r < 16; ++r) oacc[d][r] *= alpha;
            }
        }
    };

    if (ntiles > 0) {
        // ---- prologue: K(0) -> pair buffer 1 (K part); pair 0 = {K(1), V(0)} -> buffer 0; pair 1 in flight
        loadK(0);
        storeK(bufK(1));
        loadK(1);
        loadV(0);
        storeK(bufK(0));
        storeV(bufV(0));
        loadK(2);
        loadV(1);
        __syncthreads();  // #0
#pragma unroll
        for (int s = 0; s < KS; ++s) retire_loads(qf[s]);
        if (!groupB) {
            qk_tile(bufK(1));
            __syncthreads();  // #1
        } else {
            __syncthreads();  // #1  (B idles through phase 0)
            qk_tile(bufK(1));
            __syncthreads();  // #2
        }
        for (int t = 0; t < ntiles; ++t) {
            softmax_tile(t);
            __syncthreads();
            // M(t): write pair t+1 (loaded an iteration ago), prefetch pair t+2, PV(t) + QK(t+1) from pair t
            const int pb = t & 1;
            storeK(bufK(pb ^ 1));
            storeV(bufV(pb ^ 1));
            loadK(t + 3);
            loadV(t + 2);
            pv_tile(bufV(pb));
            qk_tile(bufK(pb));
            __syncthreads();
        }
        if (!groupB) __syncthreads();  // A's extra barrier: both groups execute the same number
    }

    // ---- epilogue
    char* obase = p.o + (b * p.os[0] + h * p.os[1]) * 2;
    const float l_tot = sum_across_halves(l_run);
    const float inv = l_tot > 0.f ? 1.0f / l_tot : 0.f;
    if (row < p.Sq) {
        if (p.lse != nullptr && hi == 0) {
            const float m_use = (m_run == -INFINITY) ? 0.f : m_run;
            p.lse[(int64_t)bh * p.Sq + row] = l_tot > 0.f ? (m_use + __builtin_log2f(l_tot)) * kLn2 : -INFINITY;
        }
        char* rp = obase + (int64_t)row * p.os[2] * 2;
#pragma unroll
        for (int d = 0; d < DB; ++d)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4 x;
#pragma unroll
                for (int e = 0; e < 4; ++e) x[e] = oacc[d][4 * g + e] * inv;
                typename E::vec4 y = E::cvt4(x);
                u32x2 raw;
                __builtin_memcpy(&raw, &y, 8);
                gstore8(rp + (d * 32 + 8 * g + 4 * hi) * 2, raw);
            }
    }
}

}  // namespace fasn

// fasn_fwd_pipe.h — software-pipelined forward (MODE_PLAIN / MODE_CAUSAL): same math and data layout as
// fasn_fwd_kernel.h, but each loop iteration t issues, in ONE basic block,
//     MFMA stream :  O^T += V(t-1)^T P(t-1)^T      and      S(t+1)^T = K(t+1) Q^T
//     VALU stream :  P(t)^T = exp2(c*S(t)^T - m)   (optimistic softmax_n fast path, see fasn_fwd_kernel.h)
// so a wave's own matrix instructions run under its own exponentials (an in-order wave overlaps an MFMA only with the
// instructions that follow it in program order), instead of relying on a co-resident wave being in the other phase.
// K runs two tiles ahead of V in the LDS ring: iteration t reads K(t+1) and V(t-1), writes K(t+2) and V(t), and has
// the global loads of K(t+3) and V(t+1) in flight; one s_barrier per tile.
#pragma once
#include "fasn_fwd_kernel.h"

namespace fasn {

// BURST: fetch every K / V fragment of the iteration first, then issue the 16*QB MFMAs back to back, then the exponentials
// (pinned with sched_barrier): the matrix pipe drains the burst while the VALU phase runs (tools/ubench3.cpp pattern).
template <typename Tag, int D, int QB, int MODE, int OCC, int BURST = 0>
__global__ void __launch_bounds__(256, OCC) fasn_fwd_pipe_kernel(const FwdParams p) {
    static_assert(MODE == MODE_PLAIN || MODE == MODE_CAUSAL, "masked / biased attention uses fasn_fwd_kernel");
    using E = ET<Tag>;
    using vec8 = typename E::vec8;
    constexpr int NW = 4;
    constexpr int BM = NW * QB * 32;
    constexpr int ROWB = D * 2;
    constexpr int TILEB = KT * ROWB;
    constexpr int KS = D / 16;
    constexpr int DB = D / 32;
    constexpr int CPR = D / 8;
    constexpr int NLD = (KT * CPR) / 256;

    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const ldsK = smem;              // [2][TILEB]
    char* const ldsV = smem + 2 * TILEB;  // [2][TILEB]

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
        const int ci = tid + i * 256;
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
    auto storeK = [&](int buf) {
#pragma unroll
        for (int i = 0; i < NLD; ++i) *LDS_PTR(u32x4, ldsK + buf * TILEB + ldsoff[i]) = stK[i];
    };
    auto storeV = [&](int buf) {
#pragma unroll
        for (int i = 0; i < NLD; ++i) *LDS_PTR(u32x4, ldsV + buf * TILEB + ldsoff[i]) = stV[i];
    };

    float m_run[QB], l_run[QB];
    f32x16 oacc[QB][DB];
    // two register sets, alternating per tile: set c = t&1 holds S(t) (input of this iteration) and receives P(t);
    // set c^1 holds P(t-1) (input) and receives S(t+1): no register copies between iterations
    f32x16 sacc[2][QB][2];
    vec8 pf[2][QB][2][2];
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
        const bool sink = p.n > 0.f;
        m_run[qb] = sink ? 0.f : -INFINITY;
        l_run[qb] = (sink && hi == 0) ? p.n : 0.f;
#pragma unroll
        for (int d = 0; d < DB; ++d)
#pragma unroll
            for (int r = 0; r < 16; ++r) oacc[qb][d][r] = 0.f;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
                for (int e = 0; e < 8; ++e) pf[1][qb][kb][t2][e] = 0;  // P(-1) = 0 lives in set 1 (tile 0 uses set 0)
    }

    auto qk_tile = [&](const char* tK, f32x16 (&s)[QB][2]) {  // s := K Q^T
#pragma unroll
        for (int qb = 0; qb < QB; ++qb)
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) s[qb][kb][r] = 0.f;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                vec8 kf = lds_read_rowfrag<E, D>(tK, kb * 32 + l31, ks, hi);
#pragma unroll
                for (int qb = 0; qb < QB; ++qb) s[qb][kb] = E::mfma(kf, qf[qb][ks], s[qb][kb]);
            }
    };
    auto pv_tile = [&](const char* tV, const vec8 (&pfx)[QB][2][2]) {
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
                for (int d = 0; d < DB; ++d) {
                    vec8 vf = lds_read_trfrag<E, D>(tV, kb * 32 + 16 * t2, d, lane);
#pragma unroll
                    for (int qb = 0; qb < QB; ++qb) oacc[qb][d] = E::mfma(vf, pfx[qb][kb][t2], oacc[qb][d]);
                }
    };

    // ---- prologue: V(-1) := 0 in vbuf[1]; K(0) -> kbuf[0], K(1) -> kbuf[1]; S(0); K(2), V(0) in flight
    if (ntiles > 0) {
        loadK(0);
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const u32x4 z = {0u, 0u, 0u, 0u};
            *LDS_PTR(u32x4, ldsV + TILEB + ldsoff[i]) = z;
        }
        storeK(0);
        loadK(1);
        storeK(1);
        loadK(2);
        loadV(0);
        __syncthreads();
        qk_tile(ldsK, sacc[0]);
        __syncthreads();  // every wave has read K(0) before K(2) replaces it
    }

#pragma unroll
    for (int qb = 0; qb < QB; ++qb)
#pragma unroll
        for (int s = 0; s < KS; ++s) retire_loads(qf[qb][s]);
    const int wave_first_vis = qw0 + coff;

    auto tile_body = [&](const int t, auto CSET) {
        constexpr int C = decltype(CSET)::value;
        const int k0 = t * KT;
        // the tiles loaded one iteration ago land in the buffers nobody reads during this iteration
        if (BURST < 4) {   // BURST 4 / 5: timing ablations (no staging / no staging and no barrier), results are not attention
            storeK(t & 1);   // K(t+2)
            storeV(t & 1);   // V(t)
            loadK(t + 3);
            loadV(t + 1);
        }

        bool need_mask = (k0 + KT > p.Sk);
        if (causal) need_mask = need_mask || ((k0 + KT - 1) > wave_first_vis);

        const char* tKn = ldsK + ((t + 1) & 1) * TILEB;   // K(t+1)
        const char* tVp = ldsV + ((t + 1) & 1) * TILEB;   // V(t-1)  ((t-1)&1 == (t+1)&1)
        float lnew[QB];
        bool bad = false;

        if (BURST >= 3 && !need_mask) {
            // ---- hand-ordered block (QB = 1): one group per MFMA (PV of tile t-1, then QK^T of tile t+1; 16 at D=64, 32 at D=128), the LDS
            // fragment read of the MFMA four groups later, and the softmax arithmetic of two elements per lane of tile t,
            // skewed over three groups (packed fma | two exponentials | packed add + packed convert) so no instruction
            // waits on the one in front of it. sched_barrier(0) pins the order: an in-order wave overlaps an MFMA only
            // with what follows it in program order, and the compiler's own order clusters the MFMAs.
            static_assert(BURST < 3 || QB == 1, "hand-ordered block is written for QB = 1");
            typedef float f32x2 __attribute__((ext_vector_type(2)));
            constexpr int NF = 4, LEAD = 2;
            vec8 fr[NF];
            const f32x2 c2 = {p.c, p.c}, m2 = {-m_run[0], -m_run[0]};
            f32x2 rs2 = {0.f, 0.f};
            f32x2 tq[2], xq[2];   // pk_fma results / exponentials in flight, indexed by pair parity
            // MFMA j of the block: j < NPV: PV(t-1), accumulators (d) alternating fastest; then QK^T(t+1), the two key blocks alternating
            constexpr int NPV = 4 * DB, NQK = 2 * KS, NM = NPV + NQK, VSTEP = NM / 16;   // one softmax pair every VSTEP groups
            auto ds = [&](int j) {
                if (BURST == 6 || BURST == 7) { fr[j % NF] = qf[0][j % KS]; asm volatile("" : "+v"(fr[j % NF])); return; }   // ablation: no LDS reads
                if (j < NPV) fr[j % NF] = lds_read_trfrag<E, D>(tVp, ((j / DB) >> 1) * 32 + 16 * ((j / DB) & 1), j % DB, lane);
                else fr[j % NF] = lds_read_rowfrag<E, D>(tKn, ((j - NPV) & 1) * 32 + l31, (j - NPV) >> 1, hi);
            };
            auto mm = [&](int j) {
                if (j < NPV) {
                    const int d = j % DB, kt = j / DB;   // kt = kb*2 + t2
                    oacc[0][d] = E::mfma(fr[j % NF], pf[C ^ 1][0][kt >> 1][kt & 1], oacc[0][d]);
                } else {
                    const int kb = (j - NPV) & 1, ks = (j - NPV) >> 1;
                    f32x16 z;
#pragma unroll
                    for (int r = 0; r < 16; ++r) z[r] = 0.f;
                    sacc[C ^ 1][0][kb] = E::mfma(fr[j % NF], qf[0][ks], ks == 0 ? z : sacc[C ^ 1][0][kb]);
                }
            };
            // pair i = elements (kb = i>>3, r = 2*(i&7), r+1)
            auto v_fma = [&](int i) {
                if (i < 0 || i > 15) return;
                const f32x2 s2 = {sacc[C][0][i >> 3][2 * (i & 7)], sacc[C][0][i >> 3][2 * (i & 7) + 1]};
                tq[i & 1] = __builtin_elementwise_fma(s2, c2, m2);
            };
            auto v_exp = [&](int i) {
                if (i < 0 || i > 15) return;
                xq[i & 1] = f32x2{fast_exp2(tq[i & 1][0]), fast_exp2(tq[i & 1][1])};
            };
            auto v_out = [&](int i) {
                if (i < 0 || i > 15) return;
                rs2 += xq[i & 1];
                typedef std::remove_reference_t<decltype(vec8{}[0])> el_t;
                typedef el_t el2_t __attribute__((ext_vector_type(2)));
                const el2_t h2 = __builtin_convertvector(xq[i & 1], el2_t);
                const int kb = i >> 3, t2 = (i >> 2) & 1, e = 2 * (i & 3);
                pf[C][0][kb][t2][e] = h2[0];
                pf[C][0][kb][t2][e + 1] = h2[1];
            };
#pragma unroll
            for (int j = 0; j < NF; ++j) ds(j);
            // softmax runs LEAD groups ahead of the MFMAs: it covers the latency of the first fragment reads
#pragma unroll
            for (int i = -LEAD; i < 0; ++i) {
                v_out(i + LEAD - 2);
                v_exp(i + LEAD - 1);
                v_fma(i + LEAD);
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int g = 0; g < NM; ++g) {
                mm(g);
                if (g + NF < NM) ds(g + NF);
                if (g % VSTEP == 0) {
                    const int i = g / VSTEP;
                    v_out(i + LEAD - 2);
                    v_exp(i + LEAD - 1);
                    v_fma(i + LEAD);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            v_out(15 + LEAD - 1);   // drain the skew
            v_exp(15 + LEAD);
            v_out(15 + LEAD);
            const float rs = rs2[0] + rs2[1];
            bad = !(rs <= kSumLimit);
            lnew[0] = l_run[0] + rs;
        } else if (!need_mask) {
            // ---- ONE basic block: 16*QB MFMAs (PV of the previous tile, QK^T of the next) + the exponentials of this tile
            if (BURST) {
                vec8 vfr[2][2][DB], kfr[2][KS];
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
                        for (int d = 0; d < DB; ++d) vfr[kb][t2][d] = lds_read_trfrag<E, D>(tVp, kb * 32 + 16 * t2, d, lane);
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int ks = 0; ks < KS; ++ks) kfr[kb][ks] = lds_read_rowfrag<E, D>(tKn, kb * 32 + l31, ks, hi);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
                        for (int d = 0; d < DB; ++d)
#pragma unroll
                            for (int qb = 0; qb < QB; ++qb) oacc[qb][d] = E::mfma(vfr[kb][t2][d], pf[C ^ 1][qb][kb][t2], oacc[qb][d]);
#pragma unroll
                for (int qb = 0; qb < QB; ++qb)
#pragma unroll
                    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                        for (int r = 0; r < 16; ++r) sacc[C ^ 1][qb][kb][r] = 0.f;
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                        for (int qb = 0; qb < QB; ++qb) sacc[C ^ 1][qb][kb] = E::mfma(kfr[kb][ks], qf[qb][ks], sacc[C ^ 1][qb][kb]);
                __builtin_amdgcn_sched_barrier(0);
            } else {
                pv_tile(tVp, pf[C ^ 1]);
                qk_tile(tKn, sacc[C ^ 1]);
            }
#pragma unroll
            for (int qb = 0; qb < QB; ++qb) {
                typedef float f32x2 __attribute__((ext_vector_type(2)));
                f32x2 rs2 = {0.f, 0.f};
                const f32x2 c2 = {p.c, p.c}, m2 = {-m_run[qb], -m_run[qb]};
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int t2 = 0; t2 < 2; ++t2) {
                        f32x8 x;
#pragma unroll
                        for (int e = 0; e < 8; e += 2) {
                            const f32x2 s2 = {sacc[C][qb][kb][8 * t2 + e], sacc[C][qb][kb][8 * t2 + e + 1]};
                            const f32x2 t_ = __builtin_elementwise_fma(s2, c2, m2);   // v_pk_fma_f32
                            const f32x2 pv = {fast_exp2(t_[0]), fast_exp2(t_[1])};
                            x[e] = pv[0];
                            x[e + 1] = pv[1];
                            rs2 += pv;                                                // v_pk_add_f32
                        }
                        pf[C][qb][kb][t2] = E::cvt8(x);
                    }
                const float rs = rs2[0] + rs2[1];
                bad = bad || !(rs <= kSumLimit);
                lnew[qb] = l_run[qb] + rs;
            }
            if (BURST == 2) {
                // instruction order of the block, one group per MFMA: the matrix instruction, the LDS fragment reads of a later
                // one, then this tile's softmax arithmetic for 2*QB... elements per lane (packed fma / add / convert + exponentials)
                constexpr int NM = 16 * QB;                    // MFMAs in the block
                constexpr int NDS = 16 + 2 * KS;               // ds_read_b64_tr (PV) + ds_read_b128 (QK^T)
                __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);
#pragma unroll
                for (int i = 0; i < NM; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    if (i < NDS - 8) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);
                    __builtin_amdgcn_sched_group_barrier(0x400, 2, 0);
                }
            }
        } else {
            pv_tile(tVp, pf[C ^ 1]);
            qk_tile(tKn, sacc[C ^ 1]);
        }
        if (need_mask || __any(bad)) {
            // ---- exact path: tile max, re-centre, rescale (O already contains tile t-1, which used the old max)
#pragma unroll
            for (int qb = 0; qb < QB; ++qb) {
                const int row = qw0 + qb * 32 + l31;
                const int vis = causal ? (row + coff) : 0x7fffffff;
                float mx = -INFINITY;
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int key = k0 + kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                        float y = sacc[C][qb][kb][r] * p.c;
                        const bool show = (key < p.Sk) && (key <= vis);
                        y = show ? y : -INFINITY;
                        sacc[C][qb][kb][r] = y;
                        mx = fmaxf(mx, y);
                    }
                mx = max_across_halves(mx);
                const float m_new = fmaxf(m_run[qb], mx);
                const float m_use = (m_new == -INFINITY) ? 0.f : m_new;
                const float alpha = fast_exp2(m_run[qb] - m_use);
                float rs = 0.f;
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int t2 = 0; t2 < 2; ++t2) {
                        f32x8 x;
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            x[e] = fast_exp2(sacc[C][qb][kb][8 * t2 + e] - m_use);
                            rs += x[e];
                        }
                        pf[C][qb][kb][t2] = E::cvt8(x);
                    }
                lnew[qb] = l_run[qb] * alpha + rs;
                m_run[qb] = m_new;
                if (!__all(alpha == 1.0f)) {
#pragma unroll
                    for (int d = 0; d < DB; ++d)
#pragma unroll
                        for (int r = 0; r < 16; ++r) oacc[qb][d][r] *= alpha;
                }
            }
        }
#pragma unroll
        for (int qb = 0; qb < QB; ++qb) l_run[qb] = lnew[qb];
        if (BURST != 5 && BURST != 6) __syncthreads();
    };
    for (int t = 0; t < ntiles; t += 2) {
        tile_body(t, std::integral_constant<int, 0>{});
        if (t + 1 < ntiles) tile_body(t + 1, std::integral_constant<int, 1>{});
    }
    if (ntiles > 0) {  // drain: O^T += V(last)^T P(last)^T  (P(last) sits in set (ntiles-1)&1)
        if ((ntiles - 1) & 1) pv_tile(ldsV + TILEB, pf[1]);
        else pv_tile(ldsV, pf[0]);
    }

    // ---- epilogue: O = acc / l, LSE = ln2 * (m + log2 l)
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

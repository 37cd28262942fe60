// fasn_bwd_ws256.h — backward at head dim 256 with the GEMMs of a score block split over TWO cooperating waves of one SIMD
// (the D = 256 counterparts of fasn_bwd_dkdv_ws.h / fasn_bwd_dq_ws.h, built like fasn_fwd_ws256.h: 32-row / 32-key units, LDS-DMA
// rings, one barrier per unit). Same mathematics as everywhere (fasn_bwd_kernel.h; reference: flash_attn_triton.py:146-235 with an LSE
// that carries n). Round 3 ran D = 256 on the one-wave kernels with "feature halves": two workgroups per key block each computed S
// and dP in full (9 GEMM-equivalents executed for the algorithm's 5) and spilled 87 - 156 registers.
//
//   dK/dV, a lane owns a key:   wave A:  S = Q K'^T (seeded with -LSE) -> P = exp2(S) -> P (16 bit) to LDS -> dV^T += dO^T P
//                               wave B:  dP' = dO V^T (seeded with -delta), reads P    -> dS = P o dP'      -> dK^T += Q^T dS
//   dQ, a lane owns a row:      wave A:  S^T = K Q'^T (seeded with -LSE) -> P^T = exp2(S^T) -> P^T (16 bit) to LDS
//                               wave B:  dP'^T = V dO^T (seeded with -delta), reads P^T -> dS^T = P^T o dP'^T -> dQ^T += K^T dS^T
//
// Every wave holds ONE output accumulator (128 registers) and ONE operand fragment set (64): 7 GEMM-equivalents executed, no spills.
// B runs one unit behind A; plain, causal and key-padding launches without dropout, one query head per K/V head (the rest keeps the one-wave kernels).
#pragma once
#include "fasn_bwd_kernel.h"

namespace fasn {

constexpr int B256_UNIT = 32 * 256 * 2;   // one 32-row image [32][256] 16 bit
constexpr int B256_RING = 4;
constexpr int bwd_ws256_smem_bytes() { return 2 * B256_RING * B256_UNIT + 2 * 4 * 2048 + 2 * B256_RING * 32 * 4; }

// common: a [32][256] unit straight to LDS, 512 threads x 2 sixteen-byte slots
struct Unit256Dma {
    unsigned voff[2];
    FASN_DEV void init(int tid, int64_t row_stride) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int ci = tid + i * 512;
            const int r = ci >> 5, ch = (ci & 31) ^ swz_f<256>(r);
            voff[i] = (unsigned)(r * (int)row_stride * 2 + ch * 16);
        }
    }
    FASN_DEV void dma(u32x4 rw, uint32_t unit_addr_wave, int row0, int64_t row_stride) const {
#pragma unroll
        for (int i = 0; i < 2; ++i) lds_dma16(rw, __builtin_amdgcn_readfirstlane(unit_addr_wave + i * 8192), voff[i], row0 * (int)row_stride * 2);
    }
};

// ------------------------------------------------------------------------------------------------------------------ dK, dV
// GQA = 1 (round 5; grouped-query attention, e.g. head dim 256 with 2 - 8 query heads per K/V head): one workgroup per K/V head and key block
// walks the row units of ALL query heads of its group as one long sequence - the LDS-DMA rings, the P hand-over and the barriers do not
// notice the seam between two heads, only the requests (Q / dO / statistics of head g) and the causal row index know about it - and dK / dV
// of the K/V head leave the accumulators once: no per-query-head gradient buffer, no second pass (as fasn_bwd_dkdv_ws.h does at D = 128).
template <typename Tag, int MODE, int GQA = 0>
__global__ void __launch_bounds__(512, 2) fasn_bwd_dkdv_ws256_kernel(const BwdParams bp) {
    static_assert(MODE == MODE_PLAIN || MODE == MODE_CAUSAL, "two-wave D = 256 backward: plain and causal");
    using E = ET<Tag>;
    using vec8 = typename E::vec8;
    const FwdParams& p = bp.f;
    constexpr int D = 256, KS = 16, DB = 8, BN = 128, RU = 32;
    constexpr bool causal = MODE == MODE_CAUSAL;

    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const ldsQ = smem;                                              // [RING][UNIT]
    char* const ldsDO = smem + B256_RING * B256_UNIT;                     // [RING][UNIT]
    char* const ldsP = smem + 2 * B256_RING * B256_UNIT;                  // [2][4 key blocks][2 KiB]
    float* const ldsLse = reinterpret_cast<float*>(ldsP + 2 * 4 * 2048);  // [RING][32]  -lse*log2e
    float* const ldsDlt = ldsLse + B256_RING * 32;                        // [RING][32]  -delta

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31;
    const int hi = lane >> 5;
    const int role = wave >> 2;    // 0 = A (S, P, dV), 1 = B (dP, dS, dK)
    const int kbw = wave & 3;

    const int G = GQA ? p.kvg : 1, HK = p.H / G;   // query heads per K/V head, K/V heads
    int bhk, kblk;
    block_to_work_grouped((int)blockIdx.x, p.B * HK, bp.nblk, (FASN_CAUSAL_GROUPS && causal) ? causal_head_group(p.B * HK, p.Sq * G, D) : 1, bhk, kblk);   // (causal: heads in groups, fasn_common.h)
    const int b = bhk / HK, hk = bhk % HK;
    const int h = hk * G;   // first query head of the group (the only one without GQA)
    const int kw0 = kblk * BN + kbw * 32;
    const int key = kw0 + l31;
    const int coff = p.Sk - p.Sq;
    const float* lsebase = p.lse + ((int64_t)b * p.H + h) * p.Sq;      // of head h; head h + g: + g * Sq
    const float* dltbase = bp.delta + ((int64_t)b * p.H + h) * p.Sq;

    // key-padding launches (a mask pointer next to the plain / causal kernel, fasn_bwd_d256.hip): a key block whose 128 keys are all hidden
    // has nothing to accumulate - its dK / dV rows are zeros - so the workgroup writes them and leaves before any operand is fetched
    // (round 4 walked every row unit of such a block and zeroed the result at the store)
    if (!GQA && p.mask != nullptr) {
        const bool vis = key < p.Sk && p.mask[b * p.ms[0] + h * p.ms[1] + key] != 0;
        if (!__syncthreads_or(vis ? 1 : 0)) {
            if (key < p.Sk) {
                char* rp = role == 0 ? bp.dv + (b * bp.dvs[0] + hk * bp.dvs[1] + (int64_t)key * bp.dvs[2]) * 2
                                     : bp.dk + (b * bp.dks[0] + hk * bp.dks[1] + (int64_t)key * bp.dks[2]) * 2;
#pragma unroll
                for (int c = 0; c < D / 8 / 2; ++c) gstore16(rp + (2 * c + hi) * 16, u32x4{0u, 0u, 0u, 0u});
            }
            return;
        }
    }
    const int nu_all = (p.Sq + RU - 1) / RU;
    int u0 = 0;
    if (causal) {
        const int first_row = kblk * BN - coff;
        u0 = first_row <= 0 ? 0 : first_row / RU;
    }
    const int nu1 = nu_all - u0;   // row units per query head (local index 0 .. nu1-1 = rows 32 (u0 + u) ..)
    const int nu = G * nu1;        // row units this workgroup walks: head h, then h + 1, ... (step w = g * nu1 + u)

    Unit256Dma dq_, dd_;
    dq_.init(tid, p.qs[2]);
    dd_.init(tid, bp.dos[2]);
    const char* const qhead0 = p.q + (b * p.qs[0] + h * p.qs[1]) * 2;
    const char* const dhead0 = bp.dout + (b * bp.dos[0] + h * bp.dos[1]) * 2;
    const u32x4 qrw0 = make_rsrc_words(qhead0, bp.qbytes), drw0 = make_rsrc_words(dhead0, bp.dobytes);
    const uint32_t ldsQ_w = lds_addr(ldsQ) + wave * 1024, ldsDO_w = lds_addr(ldsDO) + wave * 1024;
    // unit u: Q / dO images (2 + 2 requests per wave) and, by the first 32 threads of wave 4, its row statistics (negated: they are the
    // start values of the S / dP accumulators; rows past Sq get -inf: every weight of such a row is 0)
    float stL = 0.f, stX = 0.f;
    const int stid = tid - 256;
    // step w of the walk = unit u of query head h + g (w = g * nu1 + u); the three walkers below (requests, statistics, wave A's row
    // index) each carry their own (g, u) counters, advanced with the step they serve
    int rq_g = 0, rq_u = 0, st_g = 0, st_u = 0;
    auto requests = [&](int w) {   // (steps past the end read back as zeros: the request counts stay uniform)
        const int uu = __builtin_amdgcn_readfirstlane(w < nu ? u0 + rq_u : nu_all + 8);   // (provably uniform for the "s" operands of the DMA statements)
        if constexpr (GQA) {   // the descriptors of head h + g (scalar arithmetic, two requests per tensor and wave)
            const u32x4 qrw = make_rsrc_words(qhead0 + (int64_t)rq_g * p.qs[1] * 2, bp.qbytes), drw = make_rsrc_words(dhead0 + (int64_t)rq_g * bp.dos[1] * 2, bp.dobytes);
            dq_.dma(qrw, ldsQ_w + (w & 3) * B256_UNIT, uu * RU, p.qs[2]);
            dd_.dma(drw, ldsDO_w + (w & 3) * B256_UNIT, uu * RU, bp.dos[2]);
            if (++rq_u == nu1) { rq_u = 0; if (rq_g + 1 < G) ++rq_g; }
        } else {
            dq_.dma(qrw0, ldsQ_w + (w & 3) * B256_UNIT, uu * RU, p.qs[2]);
            dd_.dma(drw0, ldsDO_w + (w & 3) * B256_UNIT, uu * RU, bp.dos[2]);
            ++rq_u;
        }
    };
    auto stats_gload = [&](int u) {   // u: the step (its statistics slot); the row comes from this walker's own (g, u)
        if (stid >= 0 && stid < RU) {
            const int gr = (u0 + st_u) * RU + stid;
            float l = INFINITY, x = 0.f;
            if (u < nu && gr < p.Sq) {
                l = lsebase[(int64_t)st_g * p.Sq + gr];
                x = dltbase[(int64_t)st_g * p.Sq + gr];
            }
            stL = (l == -INFINITY || l == INFINITY) ? -INFINITY : -l * kLog2e;
            stX = -x;
        }
        if (++st_u == nu1) { st_u = 0; if (st_g + 1 < G) ++st_g; }   // (uniform: every thread advances the walker)
    };
    auto stats_lstore = [&](int u) {
        if (stid >= 0 && stid < RU) {
            ldsLse[(u & 3) * 32 + stid] = stL;
            ldsDlt[(u & 3) * 32 + stid] = stX;
        }
    };

    // this wave's operand fragment: K' = K * scale*log2e for A, V for B  (B operand: col = key, k = 8 features)
    vec8 opf[KS];
    {
        const bool ok = key < p.Sk;
        const char* rp = (role == 0 ? p.k + (b * p.ks[0] + hk * p.ks[1] + (int64_t)key * p.ks[2]) * 2
                                    : p.v + (b * p.vs[0] + hk * p.vs[1] + (int64_t)key * p.vs[2]) * 2) + hi * 16;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            u32x4 a = {0u, 0u, 0u, 0u};
            if (ok) a = gload16(rp + s * 32);
            __builtin_memcpy(&opf[s], &a, 16);
        }
    }
    f32x16 acc[DB];   // dV^T (A) or dK^T (B): [feature][key]
#pragma unroll
    for (int d = 0; d < DB; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[d][r] = 0.f;

    if (nu > 0) {
        // before the loop: units 0 and 1 with their statistics (iteration u requests unit u + 2)
        requests(0);
        stats_gload(0);
        stats_lstore(0);
        requests(1);
        stats_gload(1);
        stats_lstore(1);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
#pragma unroll
    for (int s = 0; s < KS; ++s) retire_loads(opf[s]);
    if (role == 0) {
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            uint16_t hk16[8];
            __builtin_memcpy(hk16, &opf[s], 16);
            f32x8 f;
#pragma unroll
            for (int e = 0; e < 8; ++e) f[e] = E::to_f32(hk16[e]) * p.c;
            opf[s] = E::cvt8(f);
        }
    }
    char* const pslot = ldsP + kbw * 2048 + lane * 16;   // + parity * 8192 (+ 1024)
    // Units above the causal diagonal of THIS wave's keys (at most three at the start of a key block's walk, the workgroup starts at
    // its first visible row) are not skipped but masked like the diagonal itself, and a wave whose keys lie past Sk computes on zero
    // operands (its rows are never stored): no skip paths, the unit bodies stay straight-line (with them hipcc spilled 113 registers).
    auto needs_mask = [&](int u) { return causal && ((u0 + u) * RU + coff) < (kw0 + 31); };
    int a_u = 0;   // wave A's unit index inside the current query head (step u of the walk = unit a_u of head u / nu1)
    auto unit_a = [&](const int u) {
        const bool need_mask = needs_mask(a_u);
        int ol = lane;   // an opaque copy of the lane id: the swizzled LDS addresses below are recomputed per unit instead of living in ~20
        asm volatile("" : "+v"(ol));   // registers across the loop next to 192 persistent ones (a spilled address comes back through scratch with a vmcnt wait)
        char* ps = pslot + (u & 1) * 8192;
        const char* tQ = ldsQ + (u & 3) * B256_UNIT;
        const char* tD = ldsDO + (u & 3) * B256_UNIT;
        const float* tL = ldsLse + (u & 3) * 32;
        const int r0 = (u0 + a_u) * RU;
        if (++a_u == nu1) a_u = 0;
        f32x16 sacc;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 a = *LDS_PTR(const f32x4, tL + 8 * g + 4 * hi);
#pragma unroll
            for (int e = 0; e < 4; ++e) sacc[4 * g + e] = a[e];
        }
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const vec8 qa = lds_read_rowfrag<E, D>(tQ, ol & 31, s, ol >> 5);
            sacc = E::mfma(qa, opf[s], sacc);
        }
        if (need_mask) {   // the causal diagonal, on the scores (rows past Sq are hidden by their -inf seeds): a small in-place pass keeps the
                           // exponential pass single (two instantiations of it cost this kernel 150 spilled registers)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = r0 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                sacc[r] = key <= row + coff ? sacc[r] : -INFINITY;
            }
        }
        vec8 pfr[2];
#pragma unroll
        for (int t2 = 0; t2 < 2; ++t2) {
            f32x8 x;
#pragma unroll
            for (int e = 0; e < 8; ++e) x[e] = fast_exp2(sacc[8 * t2 + e]);
            pfr[t2] = E::cvt8(x);
        }
        u32x4 w0, w1;
        __builtin_memcpy(&w0, &pfr[0], 16);
        __builtin_memcpy(&w1, &pfr[1], 16);
        *LDS_PTR(u32x4, ps) = w0;
        *LDS_PTR(u32x4, ps + 1024) = w1;
#pragma unroll
        for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
            for (int d = 0; d < DB; ++d) {
                const vec8 dot = lds_read_trfrag<E, D>(tD, 16 * t2, d, ol);
                acc[d] = E::mfma(dot, pfr[t2], acc[d]);
            }
    };
    auto unit_b = [&](const int u) {
        int ol = lane;
        asm volatile("" : "+v"(ol));
        const char* tQ = ldsQ + (u & 3) * B256_UNIT;
        const char* tD = ldsDO + (u & 3) * B256_UNIT;
        const float* tX = ldsDlt + (u & 3) * 32;
        const char* ps = pslot + (u & 1) * 8192;
        f32x16 pacc;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 c = *LDS_PTR(const f32x4, tX + 8 * g + 4 * hi);
#pragma unroll
            for (int e = 0; e < 4; ++e) pacc[4 * g + e] = c[e];
        }
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const vec8 da = lds_read_rowfrag<E, D>(tD, ol & 31, s, ol >> 5);
            pacc = E::mfma(da, opf[s], pacc);
        }
        __builtin_amdgcn_sched_barrier(0);   // (P is fetched behind the dP chain: 8 registers less while the fragments stream)
        const u32x4 w0 = *LDS_PTR(const u32x4, ps), w1 = *LDS_PTR(const u32x4, ps + 1024);
        vec8 dsf[2];
#pragma unroll
        for (int t2 = 0; t2 < 2; ++t2) {
            f32x8 x;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const uint32_t word = (t2 == 0 ? w0 : w1)[e >> 1];
                const float pv = E::to_f32((uint16_t)((e & 1) ? (word >> 16) : (word & 0xffffu)));
                x[e] = pv * pacc[8 * t2 + e];
            }
            dsf[t2] = E::cvt8(x);
        }
#pragma unroll
        for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
            for (int d = 0; d < DB; ++d) {
                const vec8 qt = lds_read_trfrag<E, D>(tQ, 16 * t2, d, ol);
                acc[d] = E::mfma(qt, dsf[t2], acc[d]);
            }
    };

    // iteration u: A works on unit u, B on unit u - 1, unit u + 2 is requested (ring of four: its slot held unit u - 2, released by the
    // barrier that ended iteration u - 1). Unit u + 1 - needed in iteration u + 1 - was requested in iteration u - 1: the wait that
    // ends an iteration leaves this iteration's four requests in flight. (Wave 4's statistics loads are waited for by the compiler.)
    if (nu > 0) {
        auto run = [&](auto ROLE_) {
            constexpr int ROLE = decltype(ROLE_)::value;
            for (int u = 0; u <= nu; ++u) {
                requests(u + 2);
                if (ROLE == 1) stats_gload(u + 2);
                if (ROLE == 0) {
                    if (u < nu) unit_a(u);
                } else {
                    if (u > 0) unit_b(u - 1);
                    stats_lstore(u + 2);
                }
                asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                __syncthreads();
            }
        };
        if (role == 0) run(std::integral_constant<int, 0>{});
        else run(std::integral_constant<int, 1>{});
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    if (key < p.Sk) {
        char* rp = role == 0 ? bp.dv + (b * bp.dvs[0] + hk * bp.dvs[1] + (int64_t)key * bp.dvs[2]) * 2
                             : bp.dk + (b * bp.dks[0] + hk * bp.dks[1] + (int64_t)key * bp.dks[2]) * 2;
        // key-padding launches (a mask over (batch, head, key), fasn_bwd_d256.hip) run this kernel unchanged: a hidden key's column of
        // scores touches nothing but its own dK / dV rows (LSE and delta already exclude it), so it is enough to write those as zeros
        const bool hidden = p.mask != nullptr && p.mask[b * p.ms[0] + h * p.ms[1] + key] == 0;
        const float sc = hidden ? 0.f : (role == 0 ? 1.0f : bp.scale);
        if (hidden) {   // (a hidden column may hold inf / NaN: zeros are written, not 0 * acc; both lanes of a key share the flag)
#pragma unroll
            for (int c = 0; c < D / 8 / 2; ++c) gstore16(rp + (2 * c + hi) * 16, u32x4{0u, 0u, 0u, 0u});
        } else {
#pragma unroll
            for (int d = 0; d < DB; ++d) store_block_wide<E>(rp + d * 64, acc[d], sc, hi);   // 16-byte stores (round 5, fasn_common.h)
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------ dQ
constexpr int kDq256KpTiles = 1024;   // visibility words of 64 keys kept in LDS by the key-padding instantiation (8 KiB): Sk <= 65536
constexpr int B256_IMG = 3072;            // MODE_GENERAL: one image slot of an A wave (2 KiB bias + 1 KiB mask), as in fasn_fwd_ws256.h
constexpr int bwd_dq_ws256_smem_bytes(int mode = MODE_KEYPAD) {
    return mode == MODE_GENERAL ? (B256_RING + 3) * B256_UNIT + 2 * 4 * 2048 + 4 * 2 * B256_IMG : 2 * B256_RING * B256_UNIT + 2 * 4 * 2048 + kDq256KpTiles * 8;
}

// MODE_GENERAL (round 6; a dense boolean mask and / or a 16-bit additive bias whose rows move as vectors, either may be absent; no dense dS store -
// calls that want one keep the one-wave kernel): exactly the forward's scheme (fasn_fwd_ws256.h) - wave A keeps a private ring of two
// [32 rows][32 keys] images (2 KiB bias, 1 KiB mask bytes; three LDS-DMA requests per unit, asked for two units ahead right after the unit's
// image has been read into registers) and seeds S^T with bias*log2e - LSE*log2e (-inf where the mask byte is clear). The images are paid
// for with the fourth V slot: wave B reads V one unit behind wave A's K, so V is requested one unit later at the same lead.
template <typename Tag, int MODE>
__global__ void __launch_bounds__(512, 2) fasn_bwd_dq_ws256_kernel(const BwdParams bp) {
    static_assert(MODE == MODE_PLAIN || MODE == MODE_CAUSAL || MODE == MODE_KEYPAD || MODE == MODE_GENERAL, "two-wave D = 256 backward: plain, causal, key padding, vector mask / bias");
    using E = ET<Tag>;
    using vec8 = typename E::vec8;
    const FwdParams& p = bp.f;
    constexpr int D = 256, KS = 16, DB = 8, BM = 128, KU = 32;
    constexpr bool KP = MODE == MODE_KEYPAD;   // a boolean mask over (batch, head, key), with or without the causal flag
    constexpr bool GEN = MODE == MODE_GENERAL;
    constexpr int NV = GEN ? 3 : B256_RING;    // V ring slots
    constexpr int NIMG = GEN ? 3 : 0;          // image requests of an A wave per unit
    const bool causal = MODE == MODE_CAUSAL || ((KP || GEN) && p.causal != 0);

    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const ldsK = smem;                                   // [RING][UNIT]
    char* const ldsV = smem + B256_RING * B256_UNIT;           // [NV][UNIT]
    char* const ldsP = smem + (B256_RING + NV) * B256_UNIT;    // [2][4 row blocks][2 KiB]
    uint64_t* const ldsKP = reinterpret_cast<uint64_t*>(ldsP + 2 * 4 * 2048);   // [kDq256KpTiles] (key-padding mode)
    char* const ldsI = ldsP + 2 * 4 * 2048;                    // [4 A waves][2][B256_IMG] (MODE_GENERAL)

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31;
    const int hi = lane >> 5;
    const int role = wave >> 2;   // 0 = A (S^T, P^T), 1 = B (dP^T, dS^T, dQ^T)
    const int rbw = wave & 3;

    int bh, qi;
    block_to_work_grouped((int)blockIdx.x, p.B * p.H, bp.nblk, (FASN_CAUSAL_GROUPS && causal) ? causal_head_group(p.B * p.H, p.Sk, D) : 1, bh, qi);
    const int qblk = causal ? (bp.nblk - 1 - qi) : qi;
    const int b = bh / p.H, h = bh % p.H;
    const int q0 = qblk * BM;
    const int qw0 = q0 + rbw * 32;
    const int row = qw0 + l31;
    const bool row_ok = row < p.Sq;
    const int coff = p.Sk - p.Sq;
    const int vis = causal ? row + coff : 0x7fffffff;

    int nu = (p.Sk + KU - 1) / KU;
    if (causal) {
        const int kmax = min(q0 + BM, p.Sq) - 1 + coff;
        nu = min(nu, kmax < 0 ? 0 : kmax / KU + 1);
    }

    Unit256Dma dk_, dv_;
    dk_.init(tid, p.ks[2]);
    dv_.init(tid, p.vs[2]);
    const u32x4 krw = make_rsrc_words(p.k + (b * p.ks[0] + (h / p.kvg) * p.ks[1]) * 2, p.kbytes);   // (grouped K/V: query head h reads K/V head h / kvg)
    const u32x4 vrw = make_rsrc_words(p.v + (b * p.vs[0] + (h / p.kvg) * p.vs[1]) * 2, p.vbytes);
    const uint32_t ldsK_w = lds_addr(ldsK) + wave * 1024, ldsV_w = lds_addr(ldsV) + wave * 1024;
    const int past = (p.Sk + KU - 1) / KU + 8;
    // ---- MODE_GENERAL: this A wave's images (layout and reads as in fasn_fwd_ws256.h)
    u32x4 brw = {0u, 0u, 0u, 0u}, mrw = {0u, 0u, 0u, 0u};
    unsigned bvo[2] = {0u, 0u}, mvo = 0u;
    if (GEN) {
        brw = make_rsrc_words(p.bias ? p.bias + (b * p.bs[0] + h * p.bs[1]) * 2 : p.q, p.bias ? p.bias_bytes : 0u);
        mrw = make_rsrc_words(p.mask ? reinterpret_cast<const char*>(p.mask) + (b * p.ms[0] + h * p.ms[1]) : p.q, p.mask ? p.mask_bytes : 0u);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int sl = i * 64 + lane, r = sl >> 2, c = (sl & 3) ^ swz_f<32>(r);
            bvo[i] = (unsigned)(((qw0 + r) * (int)p.bs[2] + c * 8) * 2);
        }
        mvo = (unsigned)((qw0 + (lane >> 1)) * (int)p.ms[2] + (lane & 1) * 16);
    }
    char* const img = ldsI + rbw * (2 * B256_IMG);
    const uint32_t img_a = lds_addr(img);
    auto img_dma = [&](int u, int slot) {   // 3 requests; units past the last key are out of the descriptors' range: zeros, no traffic
        const int uu = u < nu ? u : past;
#pragma unroll
        for (int i = 0; i < 2; ++i) lds_dma16(brw, __builtin_amdgcn_readfirstlane(img_a + slot * B256_IMG + i * 1024), bvo[i], (uint32_t)uu * (uint32_t)(KU * 2));
        lds_dma16(mrw, __builtin_amdgcn_readfirstlane(img_a + slot * B256_IMG + 2048), mvo, (uint32_t)uu * (uint32_t)KU);
    };
    const uint32_t nomask = (GEN && p.mask == nullptr) ? 0x01010101u : 0u;
    const char* const img_rd_b = img + l31 * 64 + hi * 8;
    const char* const img_rd_m = img + 2048 + l31 * 32 + hi * 4;
    const int img_swz = swz_f<32>(l31);
    u32x2 braw[4] = {};
    uint32_t mraw[4] = {};
    auto img_read = [&](int slot) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            braw[g] = *LDS_PTR(const u32x2, img_rd_b + slot * B256_IMG + ((g ^ img_swz) << 4));
            mraw[g] = *LDS_PTR(const uint32_t, img_rd_m + slot * B256_IMG + 8 * g);
        }
    };
    // iteration u of the loops below calls requests(u + 2): K unit u + 2 (and V unit u + 2; MODE_GENERAL: V unit u + 1, with three slots, in front of
    // the image of unit u + 2 and the K unit - vmcnt counts in order, see fasn_fwd_ws256.h)
    auto requests = [&](int u) {
        const int uu = u < nu ? u : past;
        if constexpr (GEN) {   // (the image and K requests follow in late_requests: wave A issues them once the unit's image is in its start values)
            const int uv = u - 1 < nu ? u - 1 : past;
            dv_.dma(vrw, ldsV_w + ((u - 1) % 3) * B256_UNIT, uv * KU, p.vs[2]);
        } else {
            dk_.dma(krw, ldsK_w + (u & 3) * B256_UNIT, uu * KU, p.ks[2]);
            dv_.dma(vrw, ldsV_w + (u & 3) * B256_UNIT, uu * KU, p.vs[2]);
        }
    };

    auto late_requests = [&](int u) {   // MODE_GENERAL: [A: the image of unit u into the slot whose image was just consumed] [K unit u]
        if (role == 0) img_dma(u, u & 1);
        dk_.dma(krw, ldsK_w + (u & 3) * B256_UNIT, (u < nu ? u : past) * KU, p.ks[2]);
    };

    // this wave's operand fragment (B operand: col = q row, k = 8 features): Q' for A, dO for B; and its row statistic
    vec8 opf[KS];
    float stat = 0.f;
    {
        const char* rp = (role == 0 ? p.q + (b * p.qs[0] + h * p.qs[1] + (int64_t)row * p.qs[2]) * 2
                                    : bp.dout + (b * bp.dos[0] + h * bp.dos[1] + (int64_t)row * bp.dos[2]) * 2) + hi * 16;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            u32x4 a = {0u, 0u, 0u, 0u};
            if (row_ok) a = gload16(rp + s * 32);
            __builtin_memcpy(&opf[s], &a, 16);
        }
        if (role == 0) {
            const float l = row_ok ? p.lse[(int64_t)bh * p.Sq + row] : INFINITY;
            stat = (l == -INFINITY || l == INFINITY) ? -INFINITY : -l * kLog2e;
        } else {
            stat = row_ok ? -bp.delta[(int64_t)bh * p.Sq + row] : 0.f;
        }
    }
    if (nu > 0) {
        if constexpr (GEN) {   // K units 0 and 1, V unit 0, the images of units 0 and 1
            dk_.dma(krw, ldsK_w, 0, p.ks[2]);
            dk_.dma(krw, ldsK_w + B256_UNIT, (1 < nu ? 1 : past) * KU, p.ks[2]);
            dv_.dma(vrw, ldsV_w, 0, p.vs[2]);
            if (role == 0) {
                img_dma(0, 0);
                img_dma(1, 1);
            }
        } else {
            requests(0);
            requests(1);
        }
    }
    if (KP) kp_build_words(ldsKP, p.mask ? p.mask + (b * p.ms[0] + h * p.ms[1]) : nullptr, p.Sk, (p.Sk + 63) / 64, tid, 512);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
#pragma unroll
    for (int s = 0; s < KS; ++s) retire_loads(opf[s]);
    retire_loads(stat);
    if (KP) {   // blocks behind the last visible key are not walked (every wave finds the same one; the two units requested above stay harmless)
        int last = -1;
        for (int t = lane; t < (p.Sk + 63) / 64; t += 64) {
            const uint64_t w = ldsKP[t];
            if (w != 0ull) last = 2 * t + ((w >> 32) != 0ull ? 1 : 0);
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) last = max(last, __shfl_xor(last, o));
        const int was = nu;
        nu = min(nu, __builtin_amdgcn_readfirstlane(last + 1));
        if (was > 0 && nu == 0) {   // nothing visible at all: dQ = 0
            if (role == 1 && row_ok) {
                char* rp = bp.dq + (b * bp.dqs[0] + h * bp.dqs[1] + (int64_t)row * bp.dqs[2]) * 2;
#pragma unroll
                for (int c = 0; c < D / 8; ++c)
                    if ((c & 1) == hi) gstore16(rp + c * 16, u32x4{0u, 0u, 0u, 0u});
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            return;
        }
    }
    if (role == 0) {
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            uint16_t hq[8];
            __builtin_memcpy(hq, &opf[s], 16);
            f32x8 f;
#pragma unroll
            for (int e = 0; e < 8; ++e) f[e] = E::to_f32(hq[e]) * p.c;
            opf[s] = E::cvt8(f);
        }
    }
    const int wave_first_vis = qw0 + coff, wave_last_vis = qw0 + 31 + coff;
    auto classify = [&](int u, bool& skip, bool& need_mask, uint32_t& kpb) {
        const int k0 = u * KU;
        skip = qw0 >= p.Sq;
        need_mask = k0 + KU > p.Sk;   // (a one-key K has row stride 0: its unit rows alias key 0, so keys past Sk are hidden explicitly)
        kpb = ~0u;
        if (causal) {
            skip = skip || k0 > wave_last_vis;
            need_mask = need_mask || (k0 + KU - 1) > wave_first_vis;
        }
        if (KP) {
            kpb = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(ldsKP[u >> 1] >> (32 * (u & 1))));
            skip = skip || kpb == 0u;
            need_mask = need_mask || kpb != ~0u;
        }
    };
    char* const pslot = ldsP + rbw * 2048 + lane * 16;

    auto unit_a = [&](const int u) {
        bool skip, need_mask;
        uint32_t kpb;
        classify(u, skip, need_mask, kpb);
        int ol = lane;   // an opaque copy of the lane id: the swizzled LDS addresses below are recomputed per unit instead of living in ~20
        asm volatile("" : "+v"(ol));   // registers across the loop next to 192 persistent ones (a spilled address comes back through scratch with a vmcnt wait)
        char* ps = pslot + (u & 1) * 8192;
        if (skip) {
            if constexpr (GEN) late_requests(u + 2);
            *LDS_PTR(u32x4, ps) = u32x4{0u, 0u, 0u, 0u};
            *LDS_PTR(u32x4, ps + 1024) = u32x4{0u, 0u, 0u, 0u};
            return;
        }
        const char* tK = ldsK + (u & 3) * B256_UNIT;
        f32x16 sacc;
#pragma unroll
        for (int r = 0; r < 16; ++r) sacc[r] = stat;   // (a 16-register splat kept for the whole kernel would not fit next to 192)
        if constexpr (GEN) {   // + bias*log2e, -inf where the mask byte is clear: the unit's image goes straight into the start values, then its slot is asked for again
            img_read(u & 1);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const uint32_t w = braw[r >> 2][(r & 3) >> 1];
                const float bv = __builtin_fmaf(E::to_f32((uint16_t)((r & 1) ? (w >> 16) : (w & 0xffffu))), kLog2e, stat);
                sacc[r] = (((mraw[r >> 2] | nomask) >> (8 * (r & 3))) & 0xffu) ? bv : -INFINITY;
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the image is in registers before its slot is asked for again
            late_requests(u + 2);
        }
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const vec8 kf = lds_read_rowfrag<E, D>(tK, ol & 31, s, ol >> 5);
            sacc = E::mfma(kf, opf[s], sacc);
        }
        if (need_mask) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int kin = (r & 3) + 8 * (r >> 2) + 4 * hi, key = u * KU + kin;
                sacc[r] = (key < p.Sk && key <= vis && (!KP || ((kpb >> kin) & 1u))) ? sacc[r] : -INFINITY;
            }
        }
        vec8 pfr[2];
#pragma unroll
        for (int t2 = 0; t2 < 2; ++t2) {
            f32x8 x;
#pragma unroll
            for (int e = 0; e < 8; ++e) x[e] = fast_exp2(sacc[8 * t2 + e]);
            pfr[t2] = E::cvt8(x);
        }
        u32x4 w0, w1;
        __builtin_memcpy(&w0, &pfr[0], 16);
        __builtin_memcpy(&w1, &pfr[1], 16);
        *LDS_PTR(u32x4, ps) = w0;
        *LDS_PTR(u32x4, ps + 1024) = w1;
    };
    auto unit_b = [&](const int u, f32x16 (&acc)[DB]) {
        bool skip, need_mask;
        uint32_t kpb;
        classify(u, skip, need_mask, kpb);
        int ol = lane;   // an opaque copy of the lane id: the swizzled LDS addresses below are recomputed per unit instead of living in ~20
        asm volatile("" : "+v"(ol));   // registers across the loop next to 192 persistent ones (a spilled address comes back through scratch with a vmcnt wait)
        if (skip) return;
        const char* tK = ldsK + (u & 3) * B256_UNIT;
        const char* tV = ldsV + (GEN ? u % 3 : u & 3) * B256_UNIT;
        const char* ps = pslot + (u & 1) * 8192;
        const u32x4 w0 = *LDS_PTR(const u32x4, ps), w1 = *LDS_PTR(const u32x4, ps + 1024);
        f32x16 pacc;
#pragma unroll
        for (int r = 0; r < 16; ++r) pacc[r] = stat;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const vec8 vf = lds_read_rowfrag<E, D>(tV, ol & 31, s, ol >> 5);
            pacc = E::mfma(vf, opf[s], pacc);
        }
        vec8 dsf[2];
#pragma unroll
        for (int t2 = 0; t2 < 2; ++t2) {
            f32x8 x;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const uint32_t word = (t2 == 0 ? w0 : w1)[e >> 1];
                const float pv = E::to_f32((uint16_t)((e & 1) ? (word >> 16) : (word & 0xffffu)));
                x[e] = pv * pacc[8 * t2 + e];
            }
            dsf[t2] = E::cvt8(x);
        }
#pragma unroll
        for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
            for (int d = 0; d < DB; ++d) {
                const vec8 ktf = lds_read_trfrag<E, D>(tK, 16 * t2, d, ol);
                acc[d] = E::mfma(ktf, dsf[t2], acc[d]);
            }
    };
    auto close = [&](auto ROLE_) {
        // what may stay in flight: this iteration's requests (MODE_GENERAL: 4 + the A wave's 3 image pieces)
        if constexpr (GEN && decltype(ROLE_)::value == 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 + NIMG) : "memory");
        else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        __syncthreads();
    };
    using RoleA = std::integral_constant<int, 0>;
    using RoleB = std::integral_constant<int, 1>;
    if (role == 0) {
        if (nu > 0)
            for (int u = 0; u <= nu; ++u) {
                requests(u + 2);
                if (u < nu) unit_a(u);
                else if constexpr (GEN) late_requests(u + 2);
                close(RoleA{});
            }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
        f32x16 acc[DB];   // dQ^T
#pragma unroll
        for (int d = 0; d < DB; ++d)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[d][r] = 0.f;
        if (nu > 0)
            for (int u = 0; u <= nu; ++u) {
                requests(u + 2);
                if constexpr (GEN) late_requests(u + 2);
                if (u > 0) unit_b(u - 1, acc);
                close(RoleB{});
            }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (row_ok) {
            const int lane_e = GEN ? fresh_lane_id() : lane;   // (MODE_GENERAL: the row address from a fresh lane id - the one value that was parked in scratch across the loop)
            char* rp = bp.dq + (b * bp.dqs[0] + h * bp.dqs[1] + (int64_t)(qw0 + (lane_e & 31)) * bp.dqs[2]) * 2;
#pragma unroll
            for (int d = 0; d < DB; ++d)
                store_block_wide<E>(rp + d * 64, acc[d], bp.scale, lane_e >> 5);   // 16-byte stores (round 5, fasn_common.h)
        }
    }
}

}  // namespace fasn

// fasn_bwd_pipe.h — software-pipelined backward kernels for D = 64 (plain / causal): the same mathematics and the same
// two-kernel split as fasn_bwd_kernel.h (reference: flash_attn_triton.py:146-235 differentiated with an LSE that carries n),
// but every wave runs an explicit three-stage pipeline over its 32 x 32 score blocks instead of one block at a time.
//
// Why (round 4, ISA of fasn_bwd_dkdv_kernel<bf16,64,...>): hipcc emits the block as strict phases -
//   [ds_read x10, (s_waitcnt, v_mfma) x8] [v_exp x16, ~50 VALU] [ds_read_tr x16, (s_waitcnt, v_mfma) x8]
// - every MFMA waits for an LDS read issued just in front of it and the matrix pipe idles during the element pass: 48 % of
// the wave-cycles parked at s_waitcnt, 41 % MFMA-busy with two waves per SIMD. Nothing in that stream is bound by a unit.
//
// The pipeline (block j of a wave; S = the 8 MFMAs of S and dP, E = the element pass, G = the 8 MFMAs of dV and dK):
//   phase a(j):  MFMA  S(j)      | VALU  convert P, dS of block j-1 to 16 bit   | LDS  transposed fragments of block j-1
//   phase b(j):  MFMA  G(j-1)    | VALU  p = exp2(S'), dS = p * dP' of block j  | LDS  row fragments + row statistics of block j+1
// so the operands of every MFMA were requested one phase (>= 8 MFMAs) earlier and each phase pairs 8 MFMAs with the VALU
// work of a DIFFERENT block. Two accumulator sets (even / odd block) and three Q / dO tile buffers: the one barrier per
// 64-row tile sits between a and b of the tile's second block, where it publishes tile t+1 (requested one tile earlier)
// and releases the buffer of tile t-1 for the request of tile t+2.
#pragma once
#include "fasn_bwd_kernel.h"

namespace fasn {

constexpr int PQT = 64;   // query rows per tile (two 32-row blocks)
constexpr int PNB = 3;    // Q / dO tile buffers
constexpr int pipe_dkdv_smem_bytes() { return 2 * PNB * PQT * 64 * 2 + 2 * PNB * PQT * 4; }

// dK, dV: workgroup = 4 waves x 32 keys; a lane owns a key column (S, dP, dS are [row][key] accumulator tiles).
// DROP = 1: attention-weight dropout - the keep bits of a block are drawn in its element pass (the four lanes of a key quad compute one
// (row, key quad) hash state each and exchange them with DPP quad_perm moves, see fasn_bwd_dkdv_ws.h); dP then starts at 0 and -delta
// waits in 16 registers (dropout scales dP before delta is subtracted), dV takes the kept weights and its 1/(1-p) at the end.
template <typename Tag, int MODE, int DROP = 0>
__global__ void __launch_bounds__(256, 2) fasn_bwd_dkdv_pipe_kernel(const BwdParams bp) {
    static_assert(MODE == MODE_PLAIN || MODE == MODE_CAUSAL, "pipelined dK/dV: plain and causal");
    using E = ET<Tag>;
    using vec8 = typename E::vec8;
    const FwdParams& p = bp.f;
    constexpr int D = 64, KS = 4, DB = 2, BN = 128;
    constexpr int TILEB = PQT * D * 2;   // 8 KiB
    constexpr bool causal = MODE == MODE_CAUSAL;

    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const ldsQ = smem;                                                    // [PNB][TILEB]
    char* const ldsDO = smem + PNB * TILEB;                                     // [PNB][TILEB]
    float* const ldsLse = reinterpret_cast<float*>(smem + 2 * PNB * TILEB);     // [PNB][PQT]  -lse*log2e
    float* const ldsDlt = ldsLse + PNB * PQT;                                   // [PNB][PQT]  -delta

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31;
    const int hi = lane >> 5;
    const DropSeed dsd = DROP ? drop_seed(p.seed_lo, p.seed_hi, p.rng) : DropSeed{0u, 0u};
    const DropThr dthr = drop_thr(DROP ? p.drop_thr : 1u);

    int bh, kblk0;
    block_to_work((int)blockIdx.x, p.B * p.H, (causal && p.pair) ? (bp.nblk + 1) / 2 : bp.nblk, bh, kblk0);
    const int npass = (causal && p.pair && kblk0 != bp.nblk - 1 - kblk0) ? 2 : 1;
    const int b = bh / p.H, h = bh % p.H;
    const int coff = p.Sk - p.Sq;
    const char* qbase = p.q + (b * p.qs[0] + h * p.qs[1]) * 2;
    const char* kbase = p.k + (b * p.ks[0] + h * p.ks[1]) * 2;
    const char* vbase = p.v + (b * p.vs[0] + h * p.vs[1]) * 2;
    const char* dobase = bp.dout + (b * bp.dos[0] + h * bp.dos[1]) * 2;
    const float* lsebase = p.lse + (int64_t)bh * p.Sq;
    const float* dltbase = bp.delta + (int64_t)bh * p.Sq;

    TileDma<D, 2> tdQ, tdD;
    tdQ.init(tid, p.qs[2]);
    tdD.init(tid, bp.dos[2]);
    const u32x4 qrw = make_rsrc_words(qbase, bp.qbytes), drw = make_rsrc_words(dobase, bp.dobytes);
    const uint32_t ldsQ_w = lds_addr(smem) + wave * 1024, ldsDO_w = ldsQ_w + PNB * TILEB;

    for (int pass = 0; pass < npass; ++pass) {
    if (pass) __syncthreads();   // every wave has read the last tile of the first key block before the buffers are refilled
    const int kblk = pass == 0 ? kblk0 : bp.nblk - 1 - kblk0;
    const int kw0 = kblk * BN + wave * 32;   // first key of this wave
    const int key = kw0 + l31;

    // query tiles that can see this key block
    const int ntq = (p.Sq + PQT - 1) / PQT;
    int tq0 = 0;
    if (causal) {
        const int first_row = kblk * BN - coff;
        tq0 = first_row <= 0 ? 0 : first_row / PQT;
    }
    const int nt = ntq - tq0;

    f32x16 dkacc[DB], dvacc[DB];
#pragma unroll
    for (int d = 0; d < DB; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            dkacc[d][r] = 0.f;
            dvacc[d][r] = 0.f;
        }

    if (nt > 0) {
    // K / V fragments of this wave's keys (B operand: col = key = lane&31, k = 8 contiguous features); K pre-scaled by scale*log2e
    vec8 kf[KS], vf[KS];
    {
        const bool ok = key < p.Sk;
        const char* rk = kbase + (int64_t)key * p.ks[2] * 2 + hi * 16;
        const char* rv = vbase + (int64_t)key * p.vs[2] * 2 + hi * 16;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            u32x4 a = {0u, 0u, 0u, 0u}, c = {0u, 0u, 0u, 0u};
            if (ok) {
                a = gload16(rk + s * 32);
                c = gload16(rv + s * 32);
            }
            __builtin_memcpy(&kf[s], &a, 16);
            __builtin_memcpy(&vf[s], &c, 16);
        }
    }

    float stL = 0.f, stX = 0.f;
    auto stats_gload = [&](int row0) {
        if (tid < PQT) {
            const int gr = row0 + tid;
            float l = INFINITY, x = 0.f;
            if (gr < p.Sq) {
                l = lsebase[gr];
                x = dltbase[gr];
            }
            stL = (l == -INFINITY || l == INFINITY) ? -INFINITY : -l * kLog2e;   // a row without weights: every P = exp2(-inf) = 0
            stX = -x;
        }
    };
    auto stats_lstore = [&](int buf) {
        if (tid < PQT) {
            ldsLse[buf * PQT + tid] = stL;
            ldsDlt[buf * PQT + tid] = stX;
        }
    };
    auto tile_dma = [&](int t, int buf) {   // tile t (local index) -> buffer buf
        tdQ.dma(qrw, ldsQ_w + buf * TILEB, (tq0 + t) * PQT, p.qs[2]);
        tdD.dma(drw, ldsDO_w + buf * TILEB, (tq0 + t) * PQT, bp.dos[2]);
    };

    // ---- prologue: tiles 0 and 1 requested, tile 0 published
    tile_dma(0, 0);
    stats_gload(tq0 * PQT);
    stats_lstore(0);
    if (nt > 1) {
        tile_dma(1, 1);
        stats_gload((tq0 + 1) * PQT);
    }
    if (nt > 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");   // tile 0 has landed (the four pieces of tile 1 may still fly)
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        retire_loads(kf[s]);
        retire_loads(vf[s]);
        uint16_t hk[8];
        __builtin_memcpy(hk, &kf[s], 16);
        f32x8 f;
#pragma unroll
        for (int e = 0; e < 8; ++e) f[e] = E::to_f32(hk[e]) * p.c;
        kf[s] = E::cvt8(f);
    }

    // ---- pipeline state
    f32x16 sX, pX, sY, pY;     // S' / dP' (then P / dS) of the even and the odd block in flight
    vec8 qa[KS], da[KS];       // row fragments of the block whose S MFMAs come next
    vec8 dot[2][DB], qt[2][DB];   // transposed fragments of the block whose G MFMAs come next
    vec8 pk[2], dsk[2];        // 16-bit P, dS of that block
    f32x16 xr;                 // DROP: -delta of the rows of the block whose element pass comes next
    const DropLane dlane = drop_lane(key);

    // The phases are cut into sched_barrier-delimited pieces so that the register allocator can time-share one 32-register block
    // between the row fragments (live from phase b of block j-1 to the S MFMAs of block j) and the transposed fragments (live from
    // phase a to the G MFMAs of phase b): each half is requested right behind the MFMAs that consumed the other kind.
    auto load_rf_q = [&](int bo, int qb, f32x16& s) __attribute__((always_inline)) {   // Q row fragments + S seeds (-lse*log2e)
        const char* tQ = ldsQ + bo;
        const float* tL = reinterpret_cast<const float*>(reinterpret_cast<const char*>(ldsLse) + (bo >> 5));   // bo / TILEB * PQT * 4
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) qa[ks] = lds_read_rowfrag<E, D>(tQ, qb * 32 + l31, ks, hi);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 a = *LDS_PTR(const f32x4, tL + qb * 32 + 8 * g + 4 * hi);
#pragma unroll
            for (int e = 0; e < 4; ++e) s[4 * g + e] = a[e];
        }
    };
    auto load_rf_d = [&](int bo, int qb, f32x16& pp) __attribute__((always_inline)) {   // dO row fragments + dP seeds (-delta)
        const char* tD = ldsDO + bo;
        const float* tX = reinterpret_cast<const float*>(reinterpret_cast<const char*>(ldsDlt) + (bo >> 5));
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) da[ks] = lds_read_rowfrag<E, D>(tD, qb * 32 + l31, ks, hi);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 c = *LDS_PTR(const f32x4, tX + qb * 32 + 8 * g + 4 * hi);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if (DROP) xr[4 * g + e] = c[e];
                else pp[4 * g + e] = c[e];
            }
        }
    };
    auto mfma_S1 = [&](f32x16& s) __attribute__((always_inline)) {
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) s = E::mfma(qa[ks], kf[ks], s);
    };
    auto mfma_S2 = [&](f32x16& pp) __attribute__((always_inline)) {
        if (DROP) {   // dP starts at 0
            const f32x16 zero = {};
            pp = E::mfma(da[0], vf[0], zero);
#pragma unroll
            for (int ks = 1; ks < KS; ++ks) pp = E::mfma(da[ks], vf[ks], pp);
        } else {
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) pp = E::mfma(da[ks], vf[ks], pp);
        }
    };
    auto load_tr = [&](int bo, int qb, int t2) __attribute__((always_inline)) {
        const char* tQ = ldsQ + bo;
        const char* tD = ldsDO + bo;
#pragma unroll
        for (int d = 0; d < DB; ++d) {
            dot[t2][d] = lds_read_trfrag<E, D>(tD, qb * 32 + 16 * t2, d, lane);
            qt[t2][d] = lds_read_trfrag<E, D>(tQ, qb * 32 + 16 * t2, d, lane);
        }
    };
    auto mfma_G = [&](int t2) __attribute__((always_inline)) {
#pragma unroll
        for (int d = 0; d < DB; ++d) {
            dvacc[d] = E::mfma(dot[t2][d], pk[t2], dvacc[d]);
            dkacc[d] = E::mfma(qt[t2][d], dsk[t2], dkacc[d]);
        }
    };
    auto elem_half = [&](f32x16& s, f32x16& pp, int half, int r0) __attribute__((always_inline)) {   // (r0: first row of the block)
#pragma unroll
        for (int rr = 0; rr < 8; ++rr) {
            const int r = 8 * half + rr;
            const float pv = fast_exp2(s[r]);
            if (DROP) {
                const int g = r >> 2;
                const uint32_t own = drop_mix(drop_row_base(dsd.lo, (uint32_t)bh, (uint32_t)(r0 + 8 * g + 4 * hi + (lane & 3))), dsd.hi, (uint32_t)(key >> 4));   // (one per g: CSE)
                const bool keep = (int32_t)drop_word(quad_bcast(own, r & 3), dlane) >= dthr.hi32;   // the state of THIS register's row, from the lane of the key quad that computed it
                const float ks = keep ? p.drop_scale : 0.f;                          // ONE select per weight: the factor 1 / (1 - p) or 0
                s[r] = pv * ks;                                                     // what dV multiplies
                pp[r] = pv * __builtin_fmaf(pp[r], ks, xr[r]);                       // dS = P o (dropped dP - delta)
            } else {
                s[r] = pv;
                pp[r] = pv * pp[r];
            }
        }
    };
    auto pack = [&](const f32x16& s, const f32x16& pp) __attribute__((always_inline)) {
#pragma unroll
        for (int t2 = 0; t2 < 2; ++t2) {
            f32x8 x, y;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                x[e] = s[8 * t2 + e];
                y[e] = pp[8 * t2 + e];
            }
            pk[t2] = E::cvt8(x);
            dsk[t2] = E::cvt8(y);
        }
    };
    // Visibility. Rows past Sq carry -inf seeds (stats_gload) and keys past Sk belong to lanes whose dK / dV rows are never stored, so
    // only the causal diagonal needs a per-element test - and it is applied to the SEEDS of the block (hidden score: S' = -inf, P = 0,
    // dS = 0 * dP' = 0) in a small wave-uniform branch between the phases: the phases themselves stay branch free (a branch inside
    // a phase made hipcc duplicate it and spill 300+ registers).
    auto seed_mask = [&](f32x16& s, int r0) __attribute__((always_inline)) {
        if (causal && (r0 + coff) < (kw0 + 31)) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = r0 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                s[r] = key <= row + coff ? s[r] : -INFINITY;
            }
        }
    };
    using TrueT = std::true_type;
    using FalseT = std::false_type;
#define FASN_SB() __builtin_amdgcn_sched_barrier(0)
    auto pin = [](auto& x) __attribute__((always_inline)) { asm volatile("" : "+v"(x)); };   // the value exists HERE: nothing that produces it may sink below

    // phase a of a block: S and dP of this block beside the 16-bit conversion and the transposed fragments of the previous one
    auto phase_a = [&](auto HAVE_PREV, f32x16& s, f32x16& pp, const f32x16& ps, const f32x16& ppp, int pbo, int pqb) __attribute__((always_inline)) {
        constexpr bool hp = decltype(HAVE_PREV)::value;
        mfma_S1(s);
        if (hp) {
            pack(ps, ppp);
            pin(pk[0]), pin(pk[1]), pin(dsk[0]), pin(dsk[1]);   // (IR-level sinking would move the conversion to its use behind the next branch)
        }
        FASN_SB();
        if (hp) load_tr(pbo, pqb, 0);
        mfma_S2(pp);
        FASN_SB();
        if (hp) load_tr(pbo, pqb, 1);
        FASN_SB();
    };
    // phase b of a block: G of the previous block beside the element pass of this one and the row fragments of the next
    auto phase_b = [&](auto HAVE_PREV, f32x16& s, f32x16& pp, int nbo, int nqb, int nr0, f32x16& ns, f32x16& npp) __attribute__((always_inline)) {
        constexpr bool hp = decltype(HAVE_PREV)::value;
        if (hp) mfma_G(0);
        elem_half(s, pp, 0, nr0 - 32);   // (the block in its element pass is the one before the block being requested)
        FASN_SB();
        load_rf_q(nbo, nqb, ns);
        if (hp) mfma_G(1);
        elem_half(s, pp, 1, nr0 - 32);
        pin(s), pin(pp);
        FASN_SB();
        load_rf_d(nbo, nqb, npp);
        FASN_SB();
        seed_mask(ns, nr0);   // (rows nr0 .. of the next block)
        FASN_SB();
    };

    int bo = 0;   // byte offset of tile t's buffers
    load_rf_q(0, 0, sX);
    load_rf_d(0, 0, pX);
    seed_mask(sX, tq0 * PQT);
    auto tile_body = [&](const int t, auto FIRST) __attribute__((always_inline)) {
        constexpr bool first = decltype(FIRST)::value;
        const int r0 = (tq0 + t) * PQT;
        const int bo_prev = bo == 0 ? (PNB - 1) * TILEB : bo - TILEB;
        const int bo_next = bo == (PNB - 1) * TILEB ? 0 : bo + TILEB;
        // ---- even block (rows r0 ..)
        phase_a(std::integral_constant<bool, !first>{}, sX, pX, sY, pY, bo_prev, 1);
        phase_b(std::integral_constant<bool, !first>{}, sX, pX, bo, 1, r0 + 32, sY, pY);
        // ---- odd block (rows r0 + 32 ..)
        phase_a(TrueT{}, sY, pY, sX, pX, bo, 0);
        if (t + 1 < nt) {   // publish tile t+1; the buffer of tile t-1 is free for tile t+2
            stats_lstore(bo_next == 0 ? 0 : (bo_next == TILEB ? 1 : 2));
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (t + 2 < nt) {
                tile_dma(t + 2, bo_prev == 0 ? 0 : (bo_prev == TILEB ? 1 : 2));
                stats_gload((tq0 + t + 2) * PQT);
            }
        }
        phase_b(TrueT{}, sY, pY, bo_next, 0, r0 + PQT, sX, pX);   // (after the last tile the row fragments read here are never used)
        bo = bo_next;
    };
    tile_body(0, TrueT{});
    for (int t = 1; t < nt; ++t) tile_body(t, FalseT{});
    // ---- drain: G of the last block
    {
        const int bo_last = bo == 0 ? (PNB - 1) * TILEB : bo - TILEB;
        pack(sY, pY);
        load_tr(bo_last, 1, 0);
        load_tr(bo_last, 1, 1);
        mfma_G(0);
        mfma_G(1);
    }
#undef FASN_SB
    }   // nt > 0

    char* dkbase = bp.dk + (b * bp.dks[0] + h * bp.dks[1]) * 2;
    char* dvbase = bp.dv + (b * bp.dvs[0] + h * bp.dvs[1]) * 2;
    if (key < p.Sk) {
        char* rk = dkbase + (int64_t)key * bp.dks[2] * 2;
        char* rv = dvbase + (int64_t)key * bp.dvs[2] * 2;
#pragma unroll
        for (int d = 0; d < DB; ++d) {   // 16-byte stores (round 5, store_block_wide in fasn_common.h)
            store_block_wide<E>(rk + d * 64, dkacc[d], bp.scale, hi);
            store_block_wide<E>(rv + d * 64, dvacc[d], 1.0f, hi);   // (dropout: the kept weights entered already scaled by 1 / (1 - p))
        }
    }
    }   // pass
}

// MFMA with its C / D operands in ARCHITECTURAL registers and its B operand in the accumulator half, as inline assembly: with 512
// registers per lane hipcc selects the accumulator form for EVERY builtin MFMA and copies each S / dP tile out with 16 v_accvgpr_read
// (272 moves per 64 MFMAs in the first build of the kernel below). The builtin (accumulator) form stays for dK / dV, which only
// MFMAs touch. The hazard recogniser does not look inside an asm statement: results of these MFMAs are consumed one phase (>= 8
// MFMAs) later, and mfma_guard() stands where a phase could be scheduled too close.
template <typename Tag> struct MfmaV;
template <> struct MfmaV<bf16_tag> {
    template <typename V> static FASN_DEV void first(f32x16& d, const V& a, const V& b, const f32x16& c) {
        asm("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %3" : "=&v"(d) : "v"(a), "a"(b), "v"(c));
    }
    template <typename V> static FASN_DEV void next(f32x16& d, const V& a, const V& b) {
        asm("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(d) : "v"(a), "a"(b));
    }
};
template <> struct MfmaV<f16_tag> {
    template <typename V> static FASN_DEV void first(f32x16& d, const V& a, const V& b, const f32x16& c) {
        asm("v_mfma_f32_32x32x16_f16 %0, %1, %2, %3" : "=&v"(d) : "v"(a), "a"(b), "v"(c));
    }
    template <typename V> static FASN_DEV void next(f32x16& d, const V& a, const V& b) {
        asm("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(d) : "v"(a), "a"(b));
    }
};

// ---------------------------------------------------------------------------------------------------------------------------
// The same pipeline with KB 32-key blocks per wave and ONE wave per SIMD (512 registers: the dK / dV accumulators and the K / V
// fragments, which only MFMAs touch, can live in the accumulator half of the file). Every Q / dO fragment and every row-statistics
// read from LDS then feeds KB blocks: LDS instructions per MFMA fall by KB. The row statistics are read into their own registers
// and enter as the untied C operand of the first MFMA of each chain; the causal diagonal zeroes hidden P / dS behind the element pass.
template <typename Tag, int MODE, int KB>
__global__ void __launch_bounds__(256, 1) fasn_bwd_dkdv_pipe2_kernel(const BwdParams bp) {
    static_assert(MODE == MODE_PLAIN || MODE == MODE_CAUSAL, "pipelined dK/dV: plain and causal");
    using E = ET<Tag>;
    using vec8 = typename E::vec8;
    const FwdParams& p = bp.f;
    constexpr int D = 64, KS = 4, DB = 2, BN = 4 * KB * 32;
    constexpr int TILEB = PQT * D * 2;   // 8 KiB
    constexpr bool causal = MODE == MODE_CAUSAL;

    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const ldsQ = smem;                                                    // [PNB][TILEB]
    char* const ldsDO = smem + PNB * TILEB;                                     // [PNB][TILEB]
    float* const ldsLse = reinterpret_cast<float*>(smem + 2 * PNB * TILEB);     // [PNB][PQT]  -lse*log2e
    float* const ldsDlt = ldsLse + PNB * PQT;                                   // [PNB][PQT]  -delta

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31;
    const int hi = lane >> 5;

    int bh, kblk0;
    block_to_work((int)blockIdx.x, p.B * p.H, (causal && p.pair) ? (bp.nblk + 1) / 2 : bp.nblk, bh, kblk0);
    const int npass = (causal && p.pair && kblk0 != bp.nblk - 1 - kblk0) ? 2 : 1;
    const int b = bh / p.H, h = bh % p.H;
    const int coff = p.Sk - p.Sq;
    const char* qbase = p.q + (b * p.qs[0] + h * p.qs[1]) * 2;
    const char* kbase = p.k + (b * p.ks[0] + h * p.ks[1]) * 2;
    const char* vbase = p.v + (b * p.vs[0] + h * p.vs[1]) * 2;
    const char* dobase = bp.dout + (b * bp.dos[0] + h * bp.dos[1]) * 2;
    const float* lsebase = p.lse + (int64_t)bh * p.Sq;
    const float* dltbase = bp.delta + (int64_t)bh * p.Sq;

    TileDma<D, 2> tdQ, tdD;
    tdQ.init(tid, p.qs[2]);
    tdD.init(tid, bp.dos[2]);
    const u32x4 qrw = make_rsrc_words(qbase, bp.qbytes), drw = make_rsrc_words(dobase, bp.dobytes);
    const uint32_t ldsQ_w = lds_addr(smem) + wave * 1024, ldsDO_w = ldsQ_w + PNB * TILEB;

    for (int pass = 0; pass < npass; ++pass) {
    if (pass) __syncthreads();   // every wave has read the last tile of the first key block before the buffers are refilled
    const int kblk = pass == 0 ? kblk0 : bp.nblk - 1 - kblk0;
    const int kw0 = kblk * BN + wave * (KB * 32);   // first key of this wave

    const int ntq = (p.Sq + PQT - 1) / PQT;
    int tq0 = 0;
    if (causal) {
        const int first_row = kblk * BN - coff;
        tq0 = first_row <= 0 ? 0 : first_row / PQT;
    }
    const int nt = ntq - tq0;

    f32x16 dkacc[KB][DB], dvacc[KB][DB];
#pragma unroll
    for (int kb = 0; kb < KB; ++kb)
#pragma unroll
        for (int d = 0; d < DB; ++d)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                dkacc[kb][d][r] = 0.f;
                dvacc[kb][d][r] = 0.f;
            }

    if (nt > 0) {
    vec8 kf[KB][KS], vf[KB][KS];
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) {
        const int key = kw0 + kb * 32 + l31;
        const bool ok = key < p.Sk;
        const char* rk = kbase + (int64_t)key * p.ks[2] * 2 + hi * 16;
        const char* rv = vbase + (int64_t)key * p.vs[2] * 2 + hi * 16;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            u32x4 a = {0u, 0u, 0u, 0u}, c = {0u, 0u, 0u, 0u};
            if (ok) {
                a = gload16(rk + s * 32);
                c = gload16(rv + s * 32);
            }
            __builtin_memcpy(&kf[kb][s], &a, 16);
            __builtin_memcpy(&vf[kb][s], &c, 16);
        }
    }

    float stL = 0.f, stX = 0.f;
    auto stats_gload = [&](int row0) {
        if (tid < PQT) {
            const int gr = row0 + tid;
            float l = INFINITY, x = 0.f;
            if (gr < p.Sq) {
                l = lsebase[gr];
                x = dltbase[gr];
            }
            stL = (l == -INFINITY || l == INFINITY) ? -INFINITY : -l * kLog2e;   // a row without weights: every P = exp2(-inf) = 0
            stX = -x;
        }
    };
    auto stats_lstore = [&](int buf) {
        if (tid < PQT) {
            ldsLse[buf * PQT + tid] = stL;
            ldsDlt[buf * PQT + tid] = stX;
        }
    };
    auto tile_dma = [&](int t, int buf) {   // tile t (local index) -> buffer buf
        tdQ.dma(qrw, ldsQ_w + buf * TILEB, (tq0 + t) * PQT, p.qs[2]);
        tdD.dma(drw, ldsDO_w + buf * TILEB, (tq0 + t) * PQT, bp.dos[2]);
    };

    tile_dma(0, 0);
    stats_gload(tq0 * PQT);
    stats_lstore(0);
    if (nt > 1) {
        tile_dma(1, 1);
        stats_gload((tq0 + 1) * PQT);
    }
    if (nt > 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");   // tile 0 has landed (the four pieces of tile 1 may still fly)
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
#pragma unroll
    for (int kb = 0; kb < KB; ++kb)
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            retire_loads(kf[kb][s]);
            retire_loads(vf[kb][s]);
            uint16_t hk[8];
            __builtin_memcpy(hk, &kf[kb][s], 16);
            f32x8 f;
#pragma unroll
            for (int e = 0; e < 8; ++e) f[e] = E::to_f32(hk[e]) * p.c;
            kf[kb][s] = E::cvt8(f);
        }

    f32x16 sX[KB], pX[KB], sY[KB], pY[KB];   // S' / dP' (then P / dS) of the even and the odd row block in flight
    f32x16 lr, xr;                 // row statistics of the block whose S MFMAs come next (C operand of the first MFMA of each chain)
    vec8 qa[KS], da[KS];           // row fragments of that block
    vec8 dot[2][DB], qt[2][DB];    // transposed fragments of the block whose G MFMAs come next
    vec8 pk[KB][2], dsk[KB][2];    // 16-bit P, dS of that block
#define FASN_SB() __builtin_amdgcn_sched_barrier(0)
    auto pin = [](auto& x) __attribute__((always_inline)) { asm volatile("" : "+v"(x)); };
    auto pin_a = [](auto& x) __attribute__((always_inline)) { asm volatile("" : "+a"(x)); };   // the value lives in the accumulator half of the file HERE
    auto pin_acc = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int kb = 0; kb < KB; ++kb)
#pragma unroll
            for (int d = 0; d < DB; ++d) pin_a(dkacc[kb][d]), pin_a(dvacc[kb][d]);
    };
#pragma unroll
    for (int kb = 0; kb < KB; ++kb)
#pragma unroll
        for (int s = 0; s < KS; ++s) pin_a(kf[kb][s]), pin_a(vf[kb][s]);
    pin_acc();

    auto load_rf_q = [&](int bo, int qb) __attribute__((always_inline)) {
        const char* tQ = ldsQ + bo;
        const float* tL = reinterpret_cast<const float*>(reinterpret_cast<const char*>(ldsLse) + (bo >> 5));
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) qa[ks] = lds_read_rowfrag<E, D>(tQ, qb * 32 + l31, ks, hi);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 a = *LDS_PTR(const f32x4, tL + qb * 32 + 8 * g + 4 * hi);
#pragma unroll
            for (int e = 0; e < 4; ++e) lr[4 * g + e] = a[e];
        }
    };
    auto load_rf_d = [&](int bo, int qb) __attribute__((always_inline)) {
        const char* tD = ldsDO + bo;
        const float* tX = reinterpret_cast<const float*>(reinterpret_cast<const char*>(ldsDlt) + (bo >> 5));
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) da[ks] = lds_read_rowfrag<E, D>(tD, qb * 32 + l31, ks, hi);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 c = *LDS_PTR(const f32x4, tX + qb * 32 + 8 * g + 4 * hi);
#pragma unroll
            for (int e = 0; e < 4; ++e) xr[4 * g + e] = c[e];
        }
    };
    auto mfma_S1 = [&](f32x16 (&s)[KB]) __attribute__((always_inline)) {
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) MfmaV<Tag>::first(s[kb], qa[0], kf[kb][0], lr);
#pragma unroll
        for (int ks = 1; ks < KS; ++ks)
#pragma unroll
            for (int kb = 0; kb < KB; ++kb) MfmaV<Tag>::next(s[kb], qa[ks], kf[kb][ks]);
    };
    auto mfma_S2 = [&](f32x16 (&pp)[KB]) __attribute__((always_inline)) {
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) MfmaV<Tag>::first(pp[kb], da[0], vf[kb][0], xr);
#pragma unroll
        for (int ks = 1; ks < KS; ++ks)
#pragma unroll
            for (int kb = 0; kb < KB; ++kb) MfmaV<Tag>::next(pp[kb], da[ks], vf[kb][ks]);
    };
    auto load_tr = [&](int bo, int qb, int t2) __attribute__((always_inline)) {
        const char* tQ = ldsQ + bo;
        const char* tD = ldsDO + bo;
#pragma unroll
        for (int d = 0; d < DB; ++d) {
            dot[t2][d] = lds_read_trfrag<E, D>(tD, qb * 32 + 16 * t2, d, lane);
            qt[t2][d] = lds_read_trfrag<E, D>(tQ, qb * 32 + 16 * t2, d, lane);
        }
    };
    auto mfma_G = [&](int t2) __attribute__((always_inline)) {
#pragma unroll
        for (int d = 0; d < DB; ++d)
#pragma unroll
            for (int kb = 0; kb < KB; ++kb) {
                dvacc[kb][d] = E::mfma(dot[t2][d], pk[kb][t2], dvacc[kb][d]);
                dkacc[kb][d] = E::mfma(qt[t2][d], dsk[kb][t2], dkacc[kb][d]);
            }
    };
    auto elem_half = [&](f32x16 (&s)[KB], f32x16 (&pp)[KB], int half) __attribute__((always_inline)) {
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) {
#pragma unroll
            for (int rr = 0; rr < 8; ++rr) s[kb][8 * half + rr] = fast_exp2(s[kb][8 * half + rr]);
            // dP' comes from an asm MFMA (no hazard bookkeeping by the compiler): its first reader stands behind the eight exponentials
            asm volatile("s_nop 3" : "+v"(s[kb]), "+v"(pp[kb]));
#pragma unroll
            for (int rr = 0; rr < 8; ++rr) pp[kb][8 * half + rr] *= s[kb][8 * half + rr];
        }
    };
    auto pack = [&](const f32x16 (&s)[KB], const f32x16 (&pp)[KB]) __attribute__((always_inline)) {
#pragma unroll
        for (int kb = 0; kb < KB; ++kb)
#pragma unroll
            for (int t2 = 0; t2 < 2; ++t2) {
                f32x8 x, y;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    x[e] = s[kb][8 * t2 + e];
                    y[e] = pp[kb][8 * t2 + e];
                }
                pk[kb][t2] = E::cvt8(x);
                dsk[kb][t2] = E::cvt8(y);
                pin(pk[kb][t2]), pin(dsk[kb][t2]);
            }
    };
    // causal diagonal: zero the hidden P and dS of the block (rows r0 ..) in a small wave-uniform branch between the phases
    auto diag_mask = [&](f32x16 (&s)[KB], f32x16 (&pp)[KB], int r0) __attribute__((always_inline)) {
        if (causal && (r0 + coff) < (kw0 + KB * 32 - 1)) {
#pragma unroll
            for (int kb = 0; kb < KB; ++kb) {
                const int key = kw0 + kb * 32 + l31;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = r0 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                    const bool show = key <= row + coff;
                    s[kb][r] = show ? s[kb][r] : 0.f;
                    pp[kb][r] = show ? pp[kb][r] : 0.f;
                }
            }
        }
    };
    using TrueT = std::true_type;
    using FalseT = std::false_type;
    auto phase_a = [&](auto HAVE_PREV, f32x16 (&s)[KB], f32x16 (&pp)[KB], const f32x16 (&ps)[KB], const f32x16 (&ppp)[KB], int pbo, int pqb) __attribute__((always_inline)) {
        constexpr bool hp = decltype(HAVE_PREV)::value;
        mfma_S1(s);
        if (hp) pack(ps, ppp);
        FASN_SB();
        if (hp) load_tr(pbo, pqb, 0);
        mfma_S2(pp);
        FASN_SB();
        if (hp) load_tr(pbo, pqb, 1);
        // the row statistics are the untied C operand of asm MFMAs, read over the MFMA's passes: their registers stay allocated until
        // here (the compiler believes an asm statement is done when it has issued and would hand them to the next VALU result)
        asm volatile("" ::"v"(lr), "v"(xr));
        FASN_SB();
    };
    auto phase_b = [&](auto HAVE_PREV, f32x16 (&s)[KB], f32x16 (&pp)[KB], int r0, int nbo, int nqb) __attribute__((always_inline)) {
        constexpr bool hp = decltype(HAVE_PREV)::value;
        if (hp) mfma_G(0);
        elem_half(s, pp, 0);
        FASN_SB();
        load_rf_q(nbo, nqb);
        if (hp) mfma_G(1);
        elem_half(s, pp, 1);
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) pin(s[kb]), pin(pp[kb]);
        FASN_SB();
        load_rf_d(nbo, nqb);
        FASN_SB();
        diag_mask(s, pp, r0);
        FASN_SB();
    };

    int bo = 0;
    load_rf_q(0, 0);
    load_rf_d(0, 0);
    auto tile_body = [&](const int t, auto FIRST) __attribute__((always_inline)) {
        constexpr bool first = decltype(FIRST)::value;
        const int r0 = (tq0 + t) * PQT;
        const int bo_prev = bo == 0 ? (PNB - 1) * TILEB : bo - TILEB;
        const int bo_next = bo == (PNB - 1) * TILEB ? 0 : bo + TILEB;
        phase_a(std::integral_constant<bool, !first>{}, sX, pX, sY, pY, bo_prev, 1);
        phase_b(std::integral_constant<bool, !first>{}, sX, pX, r0, bo, 1);
        phase_a(TrueT{}, sY, pY, sX, pX, bo, 0);
        if (t + 1 < nt) {
            stats_lstore(bo_next == 0 ? 0 : (bo_next == TILEB ? 1 : 2));
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (t + 2 < nt) {
                tile_dma(t + 2, bo_prev == 0 ? 0 : (bo_prev == TILEB ? 1 : 2));
                stats_gload((tq0 + t + 2) * PQT);
            }
        }
        phase_b(TrueT{}, sY, pY, r0 + 32, bo_next, 0);
        pin_acc();
        bo = bo_next;
    };
    tile_body(0, TrueT{});
    for (int t = 1; t < nt; ++t) tile_body(t, FalseT{});
    {
        const int bo_last = bo == 0 ? (PNB - 1) * TILEB : bo - TILEB;
        pack(sY, pY);
        load_tr(bo_last, 1, 0);
        load_tr(bo_last, 1, 1);
        mfma_G(0);
        mfma_G(1);
    }
#undef FASN_SB
    }   // nt > 0

    char* dkbase = bp.dk + (b * bp.dks[0] + h * bp.dks[1]) * 2;
    char* dvbase = bp.dv + (b * bp.dvs[0] + h * bp.dvs[1]) * 2;
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) {
        const int key = kw0 + kb * 32 + l31;
        if (key < p.Sk) {
            char* rk = dkbase + (int64_t)key * bp.dks[2] * 2;
            char* rv = dvbase + (int64_t)key * bp.dvs[2] * 2;
#pragma unroll
            for (int d = 0; d < DB; ++d) {
                store_block_wide<E>(rk + d * 64, dkacc[kb][d], bp.scale, hi);
                store_block_wide<E>(rv + d * 64, dvacc[kb][d], 1.0f, hi);
            }
        }
    }
    }   // pass
}

// ---------------------------------------------------------------------------------------------------------------------------
// dQ: workgroup = 4 waves x 32 query rows, walks 64-key K / V tiles; a lane owns a query row (S^T, dP^T, dS^T are [key][row]
// accumulator tiles, dS^T feeds dQ^T[d][q] += K^T[d][key] dS^T[key][q] straight from registers), as fasn_bwd_dq_kernel.
// Pipeline over the 32-key blocks j of a wave (12 MFMAs each):
//   phase A(j):  MFMA  S(j) (4)              | VALU  dS = p * dP', 16-bit conversion of block j-1   | LDS  K^T fragments of block j-1
//   phase B(j):  MFMA  dP(j) (4), G(j-1) (4) | VALU  p = exp2(S') of block j                        | LDS  K / V row fragments of block j+1
// The -LSE*log2e / -delta seeds are per-lane splats used as the untied C operand of the first MFMA of each chain. The causal
// diagonal and the ragged last key tile zero hidden P in a small wave-uniform branch between B and A.
constexpr int pipe_dq_smem_bytes() { return 2 * PNB * KT * 64 * 2; }

// DROP = 1: the keep bits are drawn with the exponentials (a lane owns a row: one hash state per key quad, as in the forward) and kept
// as the SIGN of P until the products; dP starts at 0: dS = |P| o ((kept ? dP / (1-p) : 0) - delta).
template <typename Tag, int MODE, int DROP = 0>
__global__ void __launch_bounds__(256, 2) fasn_bwd_dq_pipe_kernel(const BwdParams bp) {
    static_assert(MODE == MODE_PLAIN || MODE == MODE_CAUSAL, "pipelined dQ: plain and causal");
    using E = ET<Tag>;
    using vec8 = typename E::vec8;
    const FwdParams& p = bp.f;
    constexpr int D = 64, KS = 4, DB = 2, BM = 128;
    constexpr int TILEB = KT * D * 2;   // 8 KiB
    constexpr bool causal = MODE == MODE_CAUSAL;

    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const ldsK = smem;                   // [PNB][TILEB]
    char* const ldsV = smem + PNB * TILEB;     // [PNB][TILEB]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31;
    const int hi = lane >> 5;
    const DropSeed dsd = DROP ? drop_seed(p.seed_lo, p.seed_hi, p.rng) : DropSeed{0u, 0u};

    int bh, qi;
    block_to_work((int)blockIdx.x, p.B * p.H, (causal && p.pair) ? (bp.nblk + 1) / 2 : bp.nblk, bh, qi);
    const int npass = (causal && p.pair && qi != bp.nblk - 1 - qi) ? 2 : 1;
    const int b = bh / p.H, h = bh % p.H;
    const int coff = p.Sk - p.Sq;
    const char* qbase = p.q + (b * p.qs[0] + h * p.qs[1]) * 2;
    const char* kbase = p.k + (b * p.ks[0] + h * p.ks[1]) * 2;
    const char* vbase = p.v + (b * p.vs[0] + h * p.vs[1]) * 2;
    const char* dobase = bp.dout + (b * bp.dos[0] + h * bp.dos[1]) * 2;

    TileDma<D, 2> tdK, tdV;
    tdK.init(tid, p.ks[2]);
    tdV.init(tid, p.vs[2]);
    const u32x4 krw = make_rsrc_words(kbase, p.kbytes), vrw = make_rsrc_words(vbase, p.vbytes);
    const uint32_t ldsK_w = lds_addr(smem) + wave * 1024, ldsV_w = ldsK_w + PNB * TILEB;

    for (int pass = 0; pass < npass; ++pass) {
    if (pass) __syncthreads();   // every wave has read the last tile of the first query block before the buffers are refilled
    const int qblk = causal ? (pass == 0 ? bp.nblk - 1 - qi : qi) : qi;
    const int q0 = qblk * BM;
    const int qw0 = q0 + wave * 32;
    const int row = qw0 + l31;

    int nt = (p.Sk + KT - 1) / KT;
    if (causal) {
        const int kmax = min(q0 + BM, p.Sq) - 1 + coff;
        nt = min(nt, kmax < 0 ? 0 : (kmax / KT + 1));
    }

    f32x16 dqacc[DB];
#pragma unroll
    for (int d = 0; d < DB; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) dqacc[d][r] = 0.f;

    // delta = rowsum(O o dO) of the lane's row: computed here (row_delta, fasn_bwd_kernel.h) and published for the dK/dV kernel
    const char* const rod = p.o + (b * p.os[0] + h * p.os[1] + (int64_t)row * p.os[2]) * 2 + hi * 16;
    if (nt <= 0) {   // rows that see no key (causal, Sq > Sk): no walk, but the dK/dV kernel still reads their delta (P = 0 there: 0 x garbage must stay 0)
        u32x4 ov[KS], dv[KS];
        const bool ok = row < p.Sq;
        const char* rd = dobase + (int64_t)row * bp.dos[2] * 2 + hi * 16;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            ov[s] = dv[s] = u32x4{0u, 0u, 0u, 0u};
            if (ok) {
                ov[s] = gload16(rod + s * 32);
                dv[s] = gload16(rd + s * 32);
            }
        }
        const float dl = row_delta<E, KS>(ov, dv);
        if (ok && hi == 0) bp.delta[(int64_t)bh * p.Sq + row] = dl;
    }
    if (nt > 0) {
    // Q^T / dO^T fragments of this wave's rows (B operand: col = row = lane&31, k = 8 contiguous features); Q pre-scaled by scale*log2e
    vec8 qf[KS], dof[KS];
    u32x4 ofr[KS];   // the row's O chunks (same layout), for delta
    float lse2, dlt;
    const bool row_ok = row < p.Sq;
    {
        const bool ok = row_ok;
        const char* rq = qbase + (int64_t)row * p.qs[2] * 2 + hi * 16;
        const char* rd = dobase + (int64_t)row * bp.dos[2] * 2 + hi * 16;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            u32x4 a = {0u, 0u, 0u, 0u}, d = {0u, 0u, 0u, 0u};
            ofr[s] = u32x4{0u, 0u, 0u, 0u};
            if (ok) {
                a = gload16(rq + s * 32);
                d = gload16(rd + s * 32);
                ofr[s] = gload16(rod + s * 32);
            }
            __builtin_memcpy(&qf[s], &a, 16);
            __builtin_memcpy(&dof[s], &d, 16);
        }
        const float l = ok ? p.lse[(int64_t)bh * p.Sq + row] : 0.f;
        lse2 = (l == -INFINITY || l == INFINITY) ? INFINITY : l * kLog2e;   // a row without weights: every P = exp2(-inf) = 0
    }
    auto tile_dma = [&](int t, int buf) {
        tdK.dma(krw, ldsK_w + buf * TILEB, t * KT, p.ks[2]);
        tdV.dma(vrw, ldsV_w + buf * TILEB, t * KT, p.vs[2]);
    };
    tile_dma(0, 0);
    if (nt > 1) {
        tile_dma(1, 1);
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");   // tile 0 (and everything older) has landed
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    f32x16 sseed, dseed;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        retire_loads(qf[s]);
        retire_loads(dof[s]);
        uint16_t hq[8];
        __builtin_memcpy(hq, &qf[s], 16);
        f32x8 f;
#pragma unroll
        for (int e = 0; e < 8; ++e) f[e] = E::to_f32(hq[e]) * p.c;
        qf[s] = E::cvt8(f);
    }
    retire_loads(lse2);
    {
        u32x4 dch[KS];
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            retire_loads(ofr[s]);
            __builtin_memcpy(&dch[s], &dof[s], 16);
        }
        dlt = row_delta<E, KS>(ofr, dch);
        if (row_ok && hi == 0) bp.delta[(int64_t)bh * p.Sq + row] = dlt;
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        sseed[r] = -lse2;
        dseed[r] = -dlt;
    }

    f32x16 sX, sY, pa;          // S' (then P) of the even / odd block in flight; dP' of the current block
    vec8 kf[KS], vf[KS];        // K / V row fragments of the next block
    vec8 ktf[2][DB];            // K^T fragments of the block whose G MFMAs come next
    vec8 dsf[2];                // 16-bit dS^T of that block
#define FASN_SB() __builtin_amdgcn_sched_barrier(0)
    auto pin = [](auto& x) __attribute__((always_inline)) { asm volatile("" : "+v"(x)); };
    auto load_kf = [&](int bo, int kb) __attribute__((always_inline)) {
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) kf[ks] = lds_read_rowfrag<E, D>(ldsK + bo, kb * 32 + l31, ks, hi);
    };
    auto load_vf = [&](int bo, int kb) __attribute__((always_inline)) {
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) vf[ks] = lds_read_rowfrag<E, D>(ldsV + bo, kb * 32 + l31, ks, hi);
    };
    auto load_tr = [&](int bo, int kb) __attribute__((always_inline)) {
#pragma unroll
        for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
            for (int d = 0; d < DB; ++d) ktf[t2][d] = lds_read_trfrag<E, D>(ldsK + bo, kb * 32 + 16 * t2, d, lane);
    };
    auto mfma_S = [&](f32x16& s) __attribute__((always_inline)) {
        s = E::mfma(kf[0], qf[0], sseed);
#pragma unroll
        for (int ks = 1; ks < KS; ++ks) s = E::mfma(kf[ks], qf[ks], s);
    };
    const float ndlt = -dlt;
    const uint32_t drop_rb = DROP ? drop_row_base(dsd.lo, (uint32_t)bh, (uint32_t)row) : 0u;
    const DropThr dthr = drop_thr(DROP ? p.drop_thr : 1u);
    const uint32_t drop_rh = drop_rh_of(hi);
    auto mfma_P = [&]() __attribute__((always_inline)) {
        const f32x16 zero = {};
        pa = E::mfma(vf[0], dof[0], DROP ? zero : dseed);
#pragma unroll
        for (int ks = 1; ks < KS; ++ks) pa = E::mfma(vf[ks], dof[ks], pa);
    };
    auto mfma_G = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
            for (int d = 0; d < DB; ++d) dqacc[d] = E::mfma(ktf[t2][d], dsf[t2], dqacc[d]);
    };
    auto exps = [&](f32x16& s, int k0) __attribute__((always_inline)) {   // (k0: first key of the block)
        const DropBlock<false> db(drop_rb, dsd.hi, (uint32_t)(k0 >> 4), hi, drop_rh);   // DROP: the states of the lane's 16 weights of this block (fasn_common.h)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float pv = fast_exp2(s[r]);
            if (DROP) pv = db.keep(r, dthr) ? pv : -pv;   // (sign set = dropped: what mulpack reads)
            s[r] = pv;
        }
    };
    auto mulpack = [&](const f32x16& s) __attribute__((always_inline)) {   // dS^T = P^T o dP'^T, 16 bit
#pragma unroll
        for (int t2 = 0; t2 < 2; ++t2) {
            f32x8 y;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                if (DROP) {   // sign set = dropped weight: dS = |P| (dP * (kept ? 1 / (1 - p) : 0) - delta) - one select (the factor), one fma, one multiply
                    const float pv = s[8 * t2 + e];
                    const float ks = __builtin_signbit(pv) ? 0.f : p.drop_scale;
                    y[e] = __builtin_fabsf(pv) * __builtin_fmaf(pa[8 * t2 + e], ks, ndlt);
                } else {
                    y[e] = s[8 * t2 + e] * pa[8 * t2 + e];
                }
            }
            dsf[t2] = E::cvt8(y);
        }
    };
    const int vis = row + coff;               // last key this lane's row sees (causal)
    const int wave_first_vis = qw0 + coff;    // ... the wave's first row sees
    // zero the hidden P of a block on the causal diagonal or at the ragged end of the key range (K rows past Sk normally read back as
    // zeros and would add nothing - but a ONE-key K has row stride 0 in the ABI's broadcast convention, so its tile rows all alias key 0)
    auto diag_mask = [&](f32x16& s, int k0) __attribute__((always_inline)) {
        if ((causal && (k0 + 31) > wave_first_vis) || k0 + 32 > p.Sk) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = k0 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                s[r] = (key < p.Sk && (!causal || key <= vis)) ? s[r] : 0.f;
            }
        }
    };
    using TrueT = std::true_type;
    using FalseT = std::false_type;
    // phase A of block j: S(j) beside the products and conversion of block j-1 (ps = its P) and its K^T fragments
    auto phase_A = [&](auto HAVE_PREV, f32x16& s, f32x16& ps, int pbo, int pkb) __attribute__((always_inline)) {
        constexpr bool hp = decltype(HAVE_PREV)::value;
        mfma_S(s);
        if (hp) {
            mulpack(ps);
            pin(dsf[0]), pin(dsf[1]);
            load_tr(pbo, pkb);
        }
        FASN_SB();
    };
    // phase B of block j: dP(j) and G(j-1) beside the exponentials of block j and the row fragments of block j+1 (keys nk0 ..)
    auto phase_B = [&](auto HAVE_PREV, f32x16& s, int k0, int nbo, int nkb) __attribute__((always_inline)) {
        constexpr bool hp = decltype(HAVE_PREV)::value;
        mfma_P();
        exps(s, k0);
        if (hp) mfma_G();
        pin(s);
        FASN_SB();
        load_kf(nbo, nkb);
        load_vf(nbo, nkb);
        FASN_SB();
        diag_mask(s, k0);
        FASN_SB();
    };

    int bo = 0;   // byte offset of tile t's buffers
    load_kf(0, 0);
    load_vf(0, 0);
    auto tile_body = [&](const int t, auto FIRST) __attribute__((always_inline)) {
        constexpr bool first = decltype(FIRST)::value;
        const int k0 = t * KT;
        const int bo_prev = bo == 0 ? (PNB - 1) * TILEB : bo - TILEB;
        const int bo_next = bo == (PNB - 1) * TILEB ? 0 : bo + TILEB;
        // ---- even block (keys k0 ..)
        phase_A(std::integral_constant<bool, !first>{}, sX, sY, bo_prev, 1);
        phase_B(std::integral_constant<bool, !first>{}, sX, k0, bo, 1);
        // ---- odd block (keys k0 + 32 ..)
        phase_A(TrueT{}, sY, sX, bo, 0);
        if (t + 1 < nt) {   // publish tile t+1; the buffer of tile t-1 is free for tile t+2
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (t + 2 < nt) tile_dma(t + 2, bo_prev == 0 ? 0 : (bo_prev == TILEB ? 1 : 2));
        }
        phase_B(TrueT{}, sY, k0 + 32, bo_next, 0);   // (after the last tile the row fragments read here are never used)
        bo = bo_next;
    };
    tile_body(0, TrueT{});
    for (int t = 1; t < nt; ++t) tile_body(t, FalseT{});
    {   // drain: products, conversion and G of the last block
        const int bo_last = bo == 0 ? (PNB - 1) * TILEB : bo - TILEB;
        mulpack(sY);
        load_tr(bo_last, 1);
        mfma_G();
    }
#undef FASN_SB
    }   // nt > 0

    if (row < p.Sq) {
        char* rp = bp.dq + (b * bp.dqs[0] + h * bp.dqs[1] + (int64_t)row * bp.dqs[2]) * 2;
#pragma unroll
        for (int d = 0; d < DB; ++d) store_block_wide<E>(rp + d * 64, dqacc[d], bp.scale, hi);   // 16-byte stores (round 5, fasn_common.h)
    }
    }   // pass
}

}  // namespace fasn

// fasn_bwd_dkdv_ws.h — dK / dV with the four GEMMs of a score block split over TWO cooperating waves of one SIMD.
//
// The one-wave kernel (fasn_bwd_dkdv_kernel) keeps dK and dV accumulators, the K and V fragments and both score-shaped
// accumulators in one wave: 376 registers at D = 128, so one wave per SIMD, whose LDS latency, accumulator <-> AGPR copies and
// element pass all sit in series with its MFMAs (31 % of the MFMA peak). Here a workgroup has 8 waves for 128 keys; the waves
// w and w + 4 share a SIMD and the same 32 keys and divide the work of a [32 rows x 32 keys] block by GEMM, not by data:
//
//   wave A (w < 4):  S = Q K'^T (seeded with -LSE)  ->  P = exp2(S)  ->  P (16 bit) to LDS  ->  dV^T += dO^T P
//   wave B (w >= 4): dP' = dO V^T (seeded with -delta), reads P from LDS  ->  dS = P o dP'   ->  dK^T += Q^T dS
//
// Each wave holds ONE output accumulator and ONE operand fragment set (about 180-200 registers: two waves per SIMD, no AGPR
// traffic), each issues 16 of the block's 32 MFMAs, and the only exchange is A's packed P (2 KiB per block, lane to same
// lane: both waves hold [row][key] tiles in the same accumulator layout). B runs ONE q-tile behind A, so the P of tile t is
// published by the barrier that ends A's iteration t anyway: one barrier per q-tile, no extra synchronisation. Q / dO tiles
// live in three LDS buffers (tile t for A, t-1 for B, t+1 in flight), P in two.
// Mask / bias work exists in wave A only (B just receives zeros for hidden scores). The additive bias of a block ([32 rows] x
// the wave's [32 keys], 2 KiB) goes HBM -> LDS by two `buffer_load_dwordx4 ... lds` into a WAVE-PRIVATE ring of three slots, three
// blocks (one and a half q-tiles) ahead: private, so no barrier is involved - the wave re-requests a slot right after reading
// it - and no register is written asynchronously (the waits are hand-counted vmcnt next to the Q / dO requests). A lane owns a
// key column, so it reads its 16 rows of a block with four transposing LDS reads (`ds_read_b64_tr_b16`). A key-padding mask is
// one flag per lane, applied to the packed weights.
// Dense boolean masks (MODE_GENERAL) stay on the one-wave kernel.
#pragma once
#include "fasn_bwd_kernel.h"

namespace fasn {

#ifndef FASN_EXP_BIASHOT
#ifndef FASN_WS_FRESH_DV
#define FASN_WS_FRESH_DV 1
#endif
#define FASN_EXP_BIASHOT 0   // experiment: every bias request reads the first rows (always an L2 hit): separates fetch latency from issue cost
#endif
#ifndef FASN_WS_ATTR
#define FASN_WS_ATTR
#endif
// DROP = 1 (round 4): attention-weight dropout. Wave A draws the keep bits and publishes P with the sign bit set for a dropped weight
// (see fasn_bwd_dq_ws.h); its own dV GEMM takes the kept weights only and the 1/(1-p) goes on the dV accumulator at the end. A lane
// owns a KEY here and its registers are rows, so the (row, key quad) state the forward computes once per 4 weights would be needed once
// per weight - but the four lanes of a key quad hold the same 16 rows: each computes the state of ONE row of a 4-row register group and
// the quad exchanges them with DPP quad_perm moves (4 states instead of 16 per block and lane, the same bits as everywhere else).
template <typename Tag, int D, int MODE, int GQA = 0, int DROP = 0>   // GQA = 1: the loop over the query heads of a K/V group is compiled in
__global__ void __launch_bounds__(512, 2) FASN_WS_ATTR fasn_bwd_dkdv_ws_kernel(const BwdParams bp) {
    using E = ET<Tag>;
    using vec8 = typename E::vec8;
    const FwdParams& p = bp.f;
    constexpr int BN = 128;               // keys per workgroup: 4 key blocks x (A wave + B wave)
    constexpr int TILEB = QT * D * 2;
    constexpr int KS = D / 16;
    constexpr int DB = D / 32;
    constexpr int CPR = D / 8;
    constexpr int NLD = (QT * CPR) / 512;  // 16-byte chunks per thread per tile
    static_assert(NLD >= 1, "tile too small for 512 threads");
    constexpr int PBUF = 4 * 2 * 2048;     // one P buffer: [4 key blocks][2 row blocks][2 KiB]
    constexpr bool VBIAS = mode_has_vbias(MODE);
    constexpr bool KPD = mode_has_keypad(MODE);
    static_assert(MODE != MODE_GENERAL_SLOW && !mode_has_vmask(MODE), "element-load and dense-mask modes stay on the one-wave kernel");

    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const ldsQ = smem;                    // [3][TILEB]
    char* const ldsDO = smem + 3 * TILEB;       // [3][TILEB]
    char* const ldsP = smem + 6 * TILEB;        // [2][PBUF]
    float* const ldsLse = reinterpret_cast<float*>(smem + 6 * TILEB + 2 * PBUF);   // [3][QT]  -lse*log2e
    float* const ldsDlt = ldsLse + 3 * QT;                                          // [3][QT]  -delta
    char* const ldsBias = reinterpret_cast<char*>(ldsDlt + 3 * QT);                 // [4 A waves][3 slots][32 rows][64 B] (bias modes)

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31;
    const int hi = lane >> 5;
    const int role = wave >> 2;    // 0 = A (S, P, dV), 1 = B (dP, dS, dK)
#if FASN_PRIO_WS   // (round 5 A/B: static wave priority for one role of a SIMD's pair: 1 = wave B, 2 = wave A)
    if (role == (FASN_PRIO_WS == 1 ? 1 : 0)) __builtin_amdgcn_s_setprio(1);
#endif
    const int kbw = wave & 3;      // key block of this wave inside the workgroup's 128 keys
    const DropSeed dsd = DROP ? drop_seed(p.seed_lo, p.seed_hi, p.rng) : DropSeed{0u, 0u};
    static_assert(!(DROP && GQA), "dropout with grouped K/V stays on the one-wave kernel");

    // One workgroup per (batch, K/V head, 128-key block). Grouped-query attention (kvg query heads per K/V head): the workgroup
    // walks the q-tiles of ALL query heads of its group, one head after the other, into the same accumulators - dK / dV come out
    // per K/V head, summed in fp32 registers, and the K / V fragments are loaded once per group.
    const int kvg = GQA ? p.kvg : 1;
    const int Hkv = p.H / kvg;
    int bhk, kblk;
    if (VBIAS && p.batch_inner && (Hkv & 7) == 0) {
        // per XCD: (head, key block, batch) with the batch fastest: the B workgroups that read the same bias columns run together
        const int xcd = blockIdx.x & 7;
        int j = blockIdx.x >> 3;
        // Key-padded batch: a workgroup whose 128 keys are all padding has nothing to do, but under the in-order dispatcher its CU
        // still waits a whole round for its next turn (tools/fasn_harness timeline) - skipping 22 % of C4's key blocks bought nothing.
        // Every workgroup therefore reads the whole key-padding mask once (B x Sk bytes from L2, a few microseconds against its
        // milliseconds of work), marks the (batch, key block) pairs with a visible key, and takes the j-th VISIBLE pair of its
        // XCD's list (same order, batch fastest): the workgroups without work are the LAST ids of the launch and leave in whole rounds.
        constexpr int kCompactBytes = 64 * 1024, kCompactBlocks = 2048;
#ifdef FASN_NO_COMPACT
        const bool compact = false;
#else
        const bool compact = KPD && !GQA && p.mask != nullptr && p.ms[1] == 0 && (int64_t)p.B * p.Sk <= kCompactBytes &&
                             p.B * bp.nblk <= kCompactBlocks && (p.Sk & 15) == 0 && (p.ms[0] & 15) == 0 &&
                             (reinterpret_cast<uintptr_t>(p.mask) & 15) == 0;
#endif
        if (compact) {
            uint8_t* const vis = reinterpret_cast<uint8_t*>(smem);   // [B][nblk] (the Q buffers are not in use yet)
            for (int i = threadIdx.x; i < p.B * bp.nblk; i += 512) vis[i] = 0;
            __syncthreads();
            const int cpr = p.Sk >> 4;   // 16-byte chunks per mask row
            for (int c = threadIdx.x; c < p.B * cpr; c += 512) {
                const int bb = c / cpr, k16 = c - bb * cpr;
                const u32x4 w = *reinterpret_cast<const u32x4*>(p.mask + (int64_t)bb * p.ms[0] + 16 * k16);
                if ((w[0] | w[1] | w[2] | w[3]) != 0u) vis[bb * bp.nblk + (k16 >> 3)] = 1;   // 128 keys = 8 chunks
            }
            __syncthreads();
            int per_head = 0;
            for (int i = 0; i < p.B * bp.nblk; ++i) per_head += vis[i];
            // visible pairs first (hpx heads x per_head of them), then the pairs without a visible key: those workgroups only write
            // their zero dK / dV rows
            const int hpx = Hkv >> 3, all_pairs = p.B * bp.nblk;
            const bool want = j < hpx * per_head;
            const int per = want ? per_head : all_pairs - per_head;
            const int jj = want ? j : j - hpx * per_head;
            const int hd = jj / per;
            int r = jj - hd * per, kb = 0, bsel = -1;
            for (; kb < bp.nblk && bsel < 0; ++kb) {
                int cnt = 0;
                for (int bb = 0; bb < p.B; ++bb) cnt += (vis[bb * bp.nblk + kb] != 0) == want;
                if (r < cnt) {
                    for (int bb = 0; bb < p.B; ++bb) {
                        if (((vis[bb * bp.nblk + kb] != 0) == want) && r-- == 0) {
                            bsel = bb;
                            break;
                        }
                    }
                } else {
                    r -= cnt;
                }
            }
            __syncthreads();   // everybody has read `vis` before the first tile lands in the same LDS
            j = (hd * bp.nblk + (kb - 1)) * p.B + bsel;
        }
        const int bb = j % p.B, rest = j / p.B;
        kblk = rest % bp.nblk;
        bhk = bb * Hkv + (rest / bp.nblk) * 8 + xcd;
    } else {
        const bool causal0 = (MODE == MODE_CAUSAL) || (MODE >= MODE_GENERAL && p.causal);   // (key block 0 is the heaviest: ascending order = heaviest first)
        block_to_work_grouped(blockIdx.x, p.B * Hkv, bp.nblk, (FASN_CAUSAL_GROUPS && causal0) ? causal_head_group(p.B * Hkv, p.Sq * kvg, D) : 1, bhk, kblk);
    }
    const bool causal = (MODE == MODE_CAUSAL) || (MODE >= MODE_GENERAL && p.causal);
    const int b = bhk / Hkv, hk = bhk % Hkv;
    int h = hk * kvg;   // current query head (first of the group)
    const int kw0 = kblk * BN + kbw * 32;   // first key of this wave
    const int key = kw0 + l31;
    const int coff = p.Sk - p.Sq;
    const int bh_drop = b * p.H + h;        // (dropout instantiations: one query head per K/V head)
    const DropLane dlane = drop_lane(key);
    const DropThr dthr = drop_thr(DROP ? p.drop_thr : 1u);

    const char* kbase = p.k + (b * p.ks[0] + hk * p.ks[1]) * 2;
    const char* vbase = p.v + (b * p.vs[0] + hk * p.vs[1]) * 2;
    // per query head (set by head_setup below)
    const float* lsebase = nullptr;
    const float* dltbase = nullptr;
    u32x4 qrw = {0u, 0u, 0u, 0u}, drw = {0u, 0u, 0u, 0u}, brw = {0u, 0u, 0u, 0u};
    bool kp_keep = true, kp_none = false;

    const int ntq_all = (p.Sq + QT - 1) / QT;
    int ntq = ntq_all;
    int tq0 = 0;
    if (causal) {
        const int first_row = kblk * BN - coff;
        tq0 = first_row <= 0 ? 0 : first_row / QT;
    }

    // this wave's operand fragment: K (pre-scaled by c = scale*log2e) for A, V for B  (B operand: col = key, k = 8 features)
    vec8 opf[KS];
    {
        const bool ok = key < p.Sk;
        const char* rp = (role == 0 ? kbase + (int64_t)key * p.ks[2] * 2 : vbase + (int64_t)key * p.vs[2] * 2) + hi * 16;
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

    // ---- Q / dO tiles straight to LDS: thread `tid` owns slots tid + 512*i of a tile image
    unsigned voffQ[NLD], voffD[NLD];
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
        const int ci = tid + i * 512;
        const int row = ci / CPR, ch = (ci % CPR) ^ swz_f<D>(row);
        voffQ[i] = (unsigned)(row * (int)p.qs[2] * 2 + ch * 16);
        voffD[i] = (unsigned)(row * (int)bp.dos[2] * 2 + ch * 16);
    }
    const uint32_t ldsQ_w = lds_addr(ldsQ) + wave * 1024, ldsDO_w = lds_addr(ldsDO) + wave * 1024;
    auto tile_dma = [&](int tq, int buf) {
        const int sq = tq * QT * (int)p.qs[2] * 2, sd = tq * QT * (int)bp.dos[2] * 2;
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            lds_dma16(qrw, __builtin_amdgcn_readfirstlane(ldsQ_w + buf * TILEB + i * 8192), voffQ[i], sq);
            lds_dma16(drw, __builtin_amdgcn_readfirstlane(ldsDO_w + buf * TILEB + i * 8192), voffD[i], sd);
        }
    };
    // per-row statistics of a tile, NEGATED (they are the start values of the S / dP accumulators); rows past Sq: lse = +inf.
    // Loaded by wave 4 (a B wave): its only other vector-memory requests are the Q / dO pieces it drains at the end of every
    // iteration anyway, so the compiler's own wait for these two loads costs nothing (in an A wave it would also drain the
    // bias requests that are meant to stay in flight).
    float stL = 0.f, stX = 0.f;
    const int stid = tid - 256;
    auto stats_gload = [&](int row0) {
        if (stid >= 0 && stid < QT) {
            // (the row index from a FRESH lane id: formed from the lane id of the kernel entry, base + 4 * lane is loop invariant, the compiler keeps
            // the two 64-bit addresses live across the tile loop - and, in the grouped-K/V walk, parks them in scratch and reloads them per tile)
            const int gr = row0 + (GQA ? fresh_lane_id() + (wave - 4) * 64 : stid);   // (only where it pays: the plain instantiations answered with 4 spilled registers)
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
        if (stid >= 0 && stid < QT) {
            ldsLse[buf * QT + stid] = stL;
            ldsDlt[buf * QT + stid] = stX;
        }
    };

    // additive bias (wave A): block bq = 2*(tile - tq0) + row block lives in slot bq % 3 of the wave's ring as [32 rows][64 B]
    // (32 keys, row-major). One request = 2 x `buffer_load_dwordx4 ... lds` (16 rows x 64 bytes each; one such instruction costs about as much issue time as a dword one); rows / keys past the end
    // read 0 (range check), so requests past the last block are simply issued like the others and the counts stay uniform.
    constexpr int kBiasPieces = 2;   // vector-memory requests per block: 2 x (16 rows x 64 bytes)
    unsigned bvo = 0;
    if (VBIAS) bvo = (unsigned)(((lane >> 2) * (int)p.bs[2] + kw0 + 8 * (lane & 3)) * 2);   // 4 lanes x 16 bytes cover one row's 32 keys
    // everything that depends on the query head: Q / dO / bias descriptors, statistics, the key-padding flags (a mask may differ
    // per head). Key padding: one flag per lane for the whole head; a workgroup none of whose 128 keys is visible (the padded
    // tail of a batch element) walks no q-tile of that head.
    auto head_setup = [&]() {
        const int bh = b * p.H + h;
        qrw = make_rsrc_words(p.q + (b * p.qs[0] + h * p.qs[1]) * 2, bp.qbytes);
        drw = make_rsrc_words(bp.dout + (b * bp.dos[0] + h * bp.dos[1]) * 2, bp.dobytes);
        lsebase = p.lse + (int64_t)bh * p.Sq;
        dltbase = bp.delta + (int64_t)bh * p.Sq;
        if (VBIAS) brw = make_rsrc_words(p.bias + (b * p.bs[0] + h * p.bs[1]) * 2, p.bias_bytes);
        ntq = ntq_all;
        if (KPD) {
            kp_keep = key < p.Sk && (p.mask == nullptr || p.mask[b * p.ms[0] + h * p.ms[1] + key] != 0);   // (no mask: a bias-only call)
            kp_none = !__any(kp_keep);
            if (__syncthreads_or(kp_keep ? 1 : 0) == 0) ntq = tq0;
        }
    };
    const uint32_t ring_a = lds_addr(ldsBias) + kbw * (3 * 2048);
    auto bias_request = [&](int row0, int slot) {   // kBiasPieces vector-memory requests
#pragma unroll
        for (int i = 0; i < kBiasPieces; ++i)
            lds_dma16(brw, __builtin_amdgcn_readfirstlane(ring_a + slot * 2048 + i * 1024), bvo, (FASN_EXP_BIASHOT ? 16 * i : row0 + 16 * i) * (int)p.bs[2] * 2);
    };
    // A lane owns a key column: its 16 rows of a block come out of the row-major ring through four transposing reads (lane i of a
    // 16-lane group addresses row 4hi + i/4, keys 4(i%4).. of its half and receives rows 4hi + {0..3} of key i), as packed pairs.
    const char* const ring_rd = ldsBias + kbw * (3 * 2048) + (4 * hi + ((lane & 15) >> 2)) * 64 + (((lane >> 4) & 1) * 16 + 4 * (lane & 3)) * 2;
    // (Measured and rejected: wave B touching the rows of a later tile as an L2 prefetch - one dword per row is 64 cache lines per
    // request and costs the texture addresser more than the hidden latency is worth: C4 backward 15.5 -> 16.8 ms. With every bias
    // request forced to hit L2 (FASN_EXP_BIASHOT) the same launch takes 14.7 ms: that is all the fetch latency is worth.)
    auto bias_read = [&](int slot, u32x2 (&br)[4]) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const s16x4 t = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s16x4, ring_rd + slot * 2048 + g * 512));
            __builtin_memcpy(&br[g], &t, 8);
        }
    };
    using Set0 = std::integral_constant<int, 0>;

    auto head_issue = [&]() {   // first q-tile, its statistics and the first three bias blocks of this head
        if (tq0 < ntq) {
            tile_dma(tq0, 0);
            stats_gload(tq0 * QT);
            stats_lstore(0);
            if (VBIAS && role == 0) {
                bias_request(tq0 * QT, 0);
                bias_request(tq0 * QT + 32, 1);
                bias_request(tq0 * QT + 64, 2);
            }
        }
    };
    // one query head per K/V head (GQA = 0): the head's setup and first requests are issued here, in front of the wait for the
    // K / V fragments, outside the role branches (the per-head state then is loop-invariant for both roles)
    if (!GQA) {
        head_setup();
        head_issue();
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (!GQA) __syncthreads();
#pragma unroll
    for (int s = 0; s < KS; ++s) retire_loads(opf[s]);
    if (role == 0) {   // K' = K * scale*log2e, rounded to the operand type (like the pre-scaled q of core/flash_attn.py:81-83)
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
    auto head_prologue = [&]() {   // (grouped-query walk: per head)
        head_issue();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    };

    // wave-uniform classification of (q tile tq) x (this wave's 32 keys): identical for the A and the B wave of a key block
    auto classify = [&](int tq, bool& skip, bool& need_mask) {
        const int r0 = tq * QT;
        skip = kw0 >= p.Sk;
        need_mask = false;
        if (causal) {
            skip = skip || (r0 + QT - 1 + coff) < kw0;              // even the last row sees none of my keys
            need_mask = (r0 + coff) < (kw0 + 31);                   // the first row does not see all my keys
        }
        if (r0 + QT > p.Sq || kw0 + 32 > p.Sk) need_mask = true;
        if (KPD && kp_none) skip = true;                            // none of this wave's keys is visible to anybody
    };
    // P exchange slot of (P buffer, key block, row block): 2 KiB = lane * 16 bytes, twice
    auto pslot = [&](int pb, int qb) { return ldsP + pb * PBUF + (kbw * 2 + qb) * 2048 + lane * 16; };

    // ---- wave A: one q tile (it also requests the next tile: between its bias waits, so that their counts are exact)
    auto tile_a = [&](const int tq, auto BUF_, auto BN_, const int pb) {
        constexpr int buf = decltype(BUF_)::value;
        const int r0 = tq * QT;
        const char* tQ = ldsQ + buf * TILEB;
        const char* tD = ldsDO + buf * TILEB;
        const float* tL = ldsLse + buf * QT;
        const bool more = tq + 1 < ntq;
        bool skip, need_mask;
        classify(tq, skip, need_mask);
        // start value of the S accumulator: -lse*log2e of the register's row (+ bias*log2e). A padded key is hidden after the
        // exponential (its packed weights are cleared), not here.
        auto start = [&](int qb, const u32x2 (&br)[4], f32x16& sacc) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 a = *LDS_PTR(const f32x4, tL + qb * 32 + 8 * g + 4 * hi);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float v = a[e];
                    if (VBIAS) {
                        const uint32_t w = br[g][e >> 1];
                        v = __builtin_fmaf(E::to_f32((uint16_t)((e & 1) ? (w >> 16) : (w & 0xffffu))), kLog2e, v);
                    }
                    sacc[4 * g + e] = v;
                }
            }
        };
        auto s_gemm = [&](int qb, f32x16& sacc) {
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                const vec8 qa = lds_read_rowfrag<E, D>(tQ, qb * 32 + l31, s, hi);
                sacc = E::mfma(qa, opf[s], sacc);
            }
        };
        auto soft = [&](int qb, const f32x16& sacc, vec8 (&pfr)[2], auto MASKED) {   // P = exp2(S'), packed, and published for wave B
#pragma unroll
            for (int t2 = 0; t2 < 2; ++t2) {
                f32x8 x;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int r = 8 * t2 + e;
                    float pv = fast_exp2(sacc[r]);
                    if (decltype(MASKED)::value) {
                        const int row = r0 + qb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                        const bool show = (key < p.Sk) && (row < p.Sq) && (!causal || key <= row + coff);
                        pv = show ? pv : 0.f;
                    }
                    x[e] = pv;
                    if (DROP) {
                        // the state of (row, key quad): computed by the lane of the quad whose key & 3 equals the row's place in its
                        // 4-row register group, fetched from it with a quad_perm broadcast (r & 3 is a compile-time constant here)
                        const int g = r >> 2;
                        const uint32_t own = drop_mix(drop_row_base(dsd.lo, (uint32_t)bh_drop, (uint32_t)(r0 + qb * 32 + 8 * g + 4 * hi + (lane & 3))), dsd.hi, (uint32_t)(key >> 4));   // (one per g: CSE)
                        const bool keep = (int32_t)drop_word(quad_bcast(own, r & 3), dlane) >= dthr.hi32;
                        x[e] = keep ? pv : -pv;   // sign set = dropped (P is never negative): what wave B reads
                    }
                }
                pfr[t2] = E::cvt8(x);
            }
            char* ps = pslot(pb, qb);   // lane to same lane, 2 x 16 bytes
            u32x4 w0, w1;
            __builtin_memcpy(&w0, &pfr[0], 16);
            __builtin_memcpy(&w1, &pfr[1], 16);
            if (KPD) {   // key padding: the lane's key is hidden for every row - clear its packed weights (also what dV multiplies)
                const uint32_t kpm = kp_keep ? 0xffffffffu : 0u;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    w0[e] &= kpm;
                    w1[e] &= kpm;
                }
            }
            if (DROP) {   // the kept weights for this wave's own dV GEMM: the published pairs with the flagged halves cleared (two packed instructions per pair)
                typedef short s16x2 __attribute__((ext_vector_type(2)));
                u32x4 v0 = w0, v1 = w1;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    s16x2 h0, h1;
                    const uint32_t a0 = v0[i], a1 = v1[i];
                    __builtin_memcpy(&h0, &a0, 4);
                    __builtin_memcpy(&h1, &a1, 4);
                    h0 = h0 >> (short)15;   // 0xffff in the dropped halves
                    h1 = h1 >> (short)15;
                    uint32_t m0, m1;
                    __builtin_memcpy(&m0, &h0, 4);
                    __builtin_memcpy(&m1, &h1, 4);
                    v0[i] &= ~m0;
                    v1[i] &= ~m1;
                }
                __builtin_memcpy(&pfr[0], &v0, 16);
                __builtin_memcpy(&pfr[1], &v1, 16);
            } else if (KPD) {
                __builtin_memcpy(&pfr[0], &w0, 16);
                __builtin_memcpy(&pfr[1], &w1, 16);
            }
            *LDS_PTR(u32x4, ps) = w0;
            *LDS_PTR(u32x4, ps + 1024) = w1;
        };
        auto dv_gemm = [&](int qb, const vec8 (&pfr)[2]) {   // dV^T[d][key] += dO^T[d][q] P[q][key]
            // The addresses of the transposed dO reads from a fresh lane id (round 6) in the instantiations that otherwise keep 1 - 14 of them in scratch
            // across the tile loop (dropout, grouped K/V, fp16 bias + key padding); the others have the registers, and recomputing the addresses per
            // block costs them 2 % (config 4 backward 13.68 -> 13.97 ms)
            constexpr bool FRESH = FASN_WS_FRESH_DV && (DROP || GQA || (MODE == MODE_BIAS_KEYPAD && std::is_same<Tag, f16_tag>::value));
            const int lane = FRESH ? fresh_lane_id() : (int)(threadIdx.x & 63);
#pragma unroll
            for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
                for (int d = 0; d < DB; ++d) {
                    const vec8 dot = lds_read_trfrag<E, D>(tD, qb * 32 + 16 * t2, d, lane);
                    acc[d] = E::mfma(dot, pfr[t2], acc[d]);
                }
        };
        // Request order of this wave per tile: [bias block bq+3 (2)] [next Q / dO tile (2*NLD)] [bias block bq+4 (2)]; the wait that
        // ends an iteration leaves only the last 2 in flight, so both blocks read here landed at least one barrier ago. A slot
        // is re-requested right after its reads were issued (the data of a request arrives hundreds of cycles after the reads
        // have left the LDS queue).
        constexpr int slot0 = (2 * buf) % 3, slot1 = (2 * buf + 1) % 3;   // buf = (tq - tq0) % 3
        f32x16 sacc;
        vec8 pfr[2];
        auto block = [&](int qb) {
            s_gemm(qb, sacc);
            // (dropout instantiations: ONE copy of the element pass - the masked one; with two the compiler keeps both sets of hoisted
            // addresses and hash constants live and spills 20 - 60 registers into the tile loop)
            if (DROP || need_mask) soft(qb, sacc, pfr, std::true_type{});
            else soft(qb, sacc, pfr, std::false_type{});
            dv_gemm(qb, pfr);
        };
        // (reading a block's bias one block / one barrier early measured slower: 8 more live registers spill)
        u32x2 b0[4] = {}, b1[4] = {};
        if (VBIAS) bias_read(slot0, b0);
        start(0, b0, sacc);
        if (VBIAS) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the slot's values are in registers before it is requested again
            bias_request(r0 + QT + 32, slot0);   // block bq + 3 = second block of the next tile
        }
        if (more) tile_dma(tq + 1, decltype(BN_)::value);
        if (!skip) block(0);
        if (VBIAS) bias_read(slot1, b1);
        start(1, b1, sacc);
        if (VBIAS) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            bias_request(r0 + 2 * QT, slot1);    // block bq + 4 = first block of the tile after the next
        }
        if (!skip) block(1);
    };

    // ---- wave B: one q tile (the one wave A finished in the previous iteration); straight-line over both row blocks
    auto tile_b = [&](const int tq, auto BUF_, const int pb) {
        constexpr int buf = decltype(BUF_)::value;
        const char* tQ = ldsQ + buf * TILEB;
        const char* tD = ldsDO + buf * TILEB;
        const float* tX = ldsDlt + buf * QT;
        bool skip, need_mask;
        classify(tq, skip, need_mask);
        if (skip) return;
        f32x16 pacc[2];
        u32x4 pw[2][2];
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 c = *LDS_PTR(const f32x4, tX + qb * 32 + 8 * g + 4 * hi);
#pragma unroll
                for (int e = 0; e < 4; ++e) pacc[qb][4 * g + e] = DROP ? 0.f : c[e];   // (dropout scales dP before delta is subtracted: dP starts at 0)
            }
            const char* ps = pslot(pb, qb);
            pw[qb][0] = *LDS_PTR(const u32x4, ps);
            pw[qb][1] = *LDS_PTR(const u32x4, ps + 1024);
        }
        auto dp_gemm = [&](int qb) {
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                const vec8 da = lds_read_rowfrag<E, D>(tD, qb * 32 + l31, s, hi);
                pacc[qb] = E::mfma(da, opf[s], pacc[qb]);
            }
        };
        vec8 dsf[2][2];
        auto ds_pass = [&](int qb) {   // dS = P o (dP - delta): P arrives rounded to 16 bit (the value dV was accumulated with)
#pragma unroll
            for (int t2 = 0; t2 < 2; ++t2) {
                f32x8 x;
                f32x4 nd[2] = {};   // dropout: -delta of the rows of registers 8 t2 .. 8 t2 + 7, read again here instead of living in 32 registers
                if (DROP) {
                    nd[0] = *LDS_PTR(const f32x4, tX + qb * 32 + 8 * (2 * t2) + 4 * hi);
                    nd[1] = *LDS_PTR(const f32x4, tX + qb * 32 + 8 * (2 * t2 + 1) + 4 * hi);
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const uint32_t word = pw[qb][t2][e >> 1];
                    const float pv = E::to_f32((uint16_t)((e & 1) ? (word >> 16) : (word & 0xffffu)));
                    if (DROP) {   // sign set = dropped weight
                        const float dpe = __builtin_signbit(pv) ? 0.f : pacc[qb][8 * t2 + e] * p.drop_scale;
                        x[e] = __builtin_fabsf(pv) * (dpe + nd[e >> 2][e & 3]);
                    } else {
                        x[e] = pv * pacc[qb][8 * t2 + e];
                    }
                }
                dsf[qb][t2] = E::cvt8(x);
            }
        };
        auto dk_gemm = [&](int qb) {   // dK^T[d][key] += Q^T[d][q] dS[q][key]
#pragma unroll
            for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
                for (int d = 0; d < DB; ++d) {
                    const vec8 qt = lds_read_trfrag<E, D>(tQ, qb * 32 + 16 * t2, d, lane);
                    acc[d] = E::mfma(qt, dsf[qb][t2], acc[d]);
                }
        };
        // scheduling regions pair the MFMAs of one row block with the element pass of the other
        __builtin_amdgcn_sched_barrier(0);
        dp_gemm(0);
        __builtin_amdgcn_sched_barrier(0);
        dp_gemm(1);
        ds_pass(0);
        __builtin_amdgcn_sched_barrier(0);
        dk_gemm(0);
        ds_pass(1);
        __builtin_amdgcn_sched_barrier(0);
        dk_gemm(1);
        __builtin_amdgcn_sched_barrier(0);
    };

    // iteration t: A works on tile t, B on tile t-1, tile t+1 is in flight. Buffers relative to tq0; the loop is unrolled by the
    // three Q / dO buffers so that their offsets are compile-time constants (they fold into the ds_read immediates). Each role
    // runs its OWN copy of the loop (same trip count, same barriers): the loop-invariant LDS addresses of a role are then
    // hoisted into that role's branch only, instead of both sets staying live across one shared loop.
    using B0 = std::integral_constant<int, 0>;
    using B1 = std::integral_constant<int, 1>;
    using B2 = std::integral_constant<int, 2>;
    auto run = [&](auto ROLE_) {
        constexpr int ROLE = decltype(ROLE_)::value;
        auto body = [&](const int t, auto BA_, auto BB_, auto BN_) {
            if (ROLE == 0) {
#ifndef WS_ONLY_B
                if (t < ntq) tile_a(t, BA_, BN_, (t - tq0) & 1);   // (t == ntq: nothing left to request either)
#endif
            } else {
                if (t + 1 < ntq) {
                    tile_dma(t + 1, decltype(BN_)::value);
                    stats_gload((t + 1) * QT);
                }
#ifndef WS_ONLY_A
                if (t > tq0) tile_b(t - 1, BB_, (t - 1 - tq0) & 1);
#endif
            }
            if (ROLE == 1 && t + 1 < ntq) stats_lstore(decltype(BN_)::value);
            // tile t+1 has landed. Wave A in the bias modes leaves its newest bias request (2 pieces, issued after the tile's) in flight
            if (VBIAS && ROLE == 0 && t < ntq) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kBiasPieces) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        };
        for (int t = tq0; t <= ntq; t += 3) {
            body(t, B0{}, B2{}, B1{});
            if (t + 1 <= ntq) body(t + 1, B1{}, B0{}, B2{});
            if (t + 2 <= ntq) body(t + 2, B2{}, B1{}, B0{});
        }
    };
    // (the loop over the group's heads sits INSIDE each role's branch: hoisted per-role addresses stay in their branch)
    auto all_heads = [&](auto ROLE_) {
        for (int g = 0; g < kvg; ++g) {
            h = hk * kvg + g;
            head_setup();
            head_prologue();
            if (tq0 < ntq) run(ROLE_);
        }
    };
    if (GQA) {
        if (role == 0) all_heads(std::integral_constant<int, 0>{});
        else all_heads(std::integral_constant<int, 1>{});
    } else if (tq0 < ntq) {
        if (role == 0) run(std::integral_constant<int, 0>{});
        else run(std::integral_constant<int, 1>{});
    }

    // ---- epilogue: A writes dV, B writes dK * scale (the output row address from a fresh lane id: nothing of it is live across the tile loop)
    const int lane_e = GQA ? fresh_lane_id() : lane;
    const int key_e = kw0 + (lane_e & 31), hi_e = lane_e >> 5;
    if (key_e < p.Sk) {
        char* rp = role == 0 ? bp.dv + (b * bp.dvs[0] + hk * bp.dvs[1] + (int64_t)key_e * bp.dvs[2]) * 2     // dK / dV are [B, H / kvg, Sk, D]
                             : bp.dk + (b * bp.dks[0] + hk * bp.dks[1] + (int64_t)key_e * bp.dks[2]) * 2;
        const float sc = role == 0 ? (DROP ? p.drop_scale : 1.0f) : bp.scale;
#pragma unroll
        for (int d = 0; d < DB; ++d) store_block_narrow<E>(rp + d * 64, acc[d], sc, hi_e);   // (8-byte stores: at its register limit, fasn_common.h)
    }
}


}  // namespace fasn

// fasn_bwd_dbias.h — gradient of an additive bias that BROADCASTS over the batch and / or the heads, reduced in the kernel.
//
// dbias[bb,hb,i,j] = sum over the (b,h) that read bias[bb,hb,i,j] of dS[b,h,i,j],   dS = P o (dP - delta)
// (the reference differentiates its additive mask through SDPA, core/flash_attn.py:100-113; an ALiBi-style [H,L,S] bias is
// read by every batch element). The dQ kernels can store dS densely, [B,H,L,S], for autograd to sum - 17 GB at
// (4,32,8192,128) for a 4.3 GB bias, written once and read once more by the reduction. Here one workgroup owns a
// [128 rows x 128 keys] tile of ONE bias slice and walks the (b,h) that share it: S^T and dP^T are recomputed per (b,h)
// (2 GEMM-equivalents on top of the backward's 7, lane = query row as in the dQ kernel), dS is summed in fp32 registers and
// stored once in the bias's own dtype and layout. No [B,H,L,S] buffer, no atomics, deterministic.
//
// Memory access. A lane owns a query ROW, so anything it loads or stores per row directly (its 32 bytes of a bias row, its Q / dO
// fragments, its 32 bytes of dbias) is 32 - 64 cache lines per wave instruction - the first version spent 90 % of its time there.
// Everything row-shaped therefore goes through LDS images filled / drained with coalesced 16-byte pieces: the bias tile once
// per workgroup (it is the same for every (b,h) the workgroup walks - that is what "broadcast" means), the Q / dO rows once per
// (b,h) (requested while the previous one is being worked on), the summed dS once at the end. K / V tiles of 64 keys come by
// LDS-DMA, double buffered over the flat sequence of steps (b, h, tile), key-permuted as in the vector mask / bias kernels so
// that a lane's 16 accumulator registers of a 32-key block are 16 CONSECUTIVE keys.
#pragma once
#include "fasn_bwd_kernel.h"

namespace fasn {

struct DbiasParams {
    BwdParams b;           // q,k,v,lse,dout,delta,mask,bias (+ strides, sizes, scale, causal); b.dbias = output base
    int Bb, Hb;            // extent of the bias over batch / heads: 1 = broadcast (reduce over it), else B / H
    int out_f32;           // dbias elements are fp32 (else the 16-bit type of q)
    int nqb, nkb;          // 128-row / 128-key blocks
};

constexpr int dbias_smem_bytes(int D) { return 4 * KT * D * 2 + (D <= 128 ? 65536 : 32768); }   // K / V tiles + the staging area

// FAST: 16-bit bias and dbias with 16-byte-movable rows and a mask (if any) with 4-byte-movable rows - the instantiation without any
// per-element global access (its 64-bit stride arithmetic and the scalar registers it pins cost the generic one 3 x the time)
template <typename Tag, int D, bool FAST>
__global__ void __launch_bounds__(256, 1) fasn_bwd_dbias_kernel(const DbiasParams dp) {
    using E = ET<Tag>;
    using vec8 = typename E::vec8;
    const BwdParams& bp = dp.b;
    const FwdParams& p = bp.f;
    constexpr int KS = D / 16;
    constexpr int TILEB = KT * D * 2;
    constexpr int NLD = (KT * (D / 8)) / 256;
    constexpr bool STAGE = D <= 128;                          // Q / dO rows through LDS (D = 256: no room, register loads)
    constexpr int ROWSB = 128 * D * 2;                        // one [128 rows][D] image
    constexpr int NLDR = (128 * (D / 8)) / 256;               // its 16-byte chunks per thread
    constexpr int STGB = STAGE ? 65536 : 32768;               // staging area: bias tile, then Q / dO rows, then the dS tile

    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const ldsK = smem;                   // [2][TILEB]
    char* const ldsV = smem + 2 * TILEB;       // [2][TILEB]
    char* const ldsS = smem + 4 * TILEB;       // [STGB]
    char* const ldsQ = ldsS;                   // [ROWSB]  (STAGE)
    char* const ldsDO = ldsS + ROWSB;          // [ROWSB]  (STAGE; 2 * ROWSB <= 64 KiB)

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31;
    const int hi = lane >> 5;

    // persistent workgroups: a tile is ~10 us of work, about what it costs to start a 4-wave workgroup that owns 128 KiB of LDS and
    // the whole register file - so each workgroup walks tiles with the stride of the grid instead of ending after one
    const int ntiles = dp.Bb * dp.Hb * dp.nqb * dp.nkb;
    for (int tile_id = blockIdx.x; tile_id < ntiles; tile_id += gridDim.x) {
    int blk = tile_id;
    const int kblk = blk % dp.nkb; blk /= dp.nkb;
    const int qblk = blk % dp.nqb; blk /= dp.nqb;
    const int hb = blk % dp.Hb;
    const int bb = blk / dp.Hb;
    const int lrow = wave * 32 + l31;          // this lane's row inside the workgroup's 128
    const int row = qblk * 128 + lrow;
    const bool row_ok = row < p.Sq;
    const int key0 = kblk * 128;
    const int coff = p.Sk - p.Sq;
    const bool causal = p.causal != 0;
    const int vis = causal ? row + coff : 0x7fffffff;

    f32x16 dsum[2][2];   // [64-key tile][32-key block]: keys key0 + 64 t + 32 kb + 16 hi + r
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) dsum[t][kb][r] = 0.f;

    // 64-key tiles of this block with a key some row of the workgroup can see (the others stay zero)
    const int last_vis = causal ? min(qblk * 128 + 127, p.Sq - 1) + coff : 0x7fffffff;
    const int nt = (key0 >= p.Sk || key0 > last_vis) ? 0 : ((key0 + KT >= p.Sk || key0 + KT > last_vis) ? 1 : 2);
    const int b_lo = dp.Bb == 1 ? 0 : bb, nb = dp.Bb == 1 ? p.B : 1;
    const int h_lo = dp.Hb == 1 ? 0 : hb, nh = dp.Hb == 1 ? p.H : 1;
    const int nsteps = nb * nh * nt;

    // ---- the bias / dbias tile image in the staging area: [128 rows][128 keys] of 2- or 4-byte elements, rows of 256 / 512 bytes
    // with their 16-byte chunks XOR-permuted like a D = 128 / 256 K/V tile (conflict free for "row = lane, same chunk" reads).
    // Usable when the tile fits the area and the global rows can be moved in 16-byte pieces.
    const int besz = FAST ? 2 : (p.bias_f32 ? 4 : 2), oesz = FAST ? 2 : (dp.out_f32 ? 4 : 2);
    const bool bias_img = FAST ? nsteps > 0 : (p.bias_vec != 0 && p.bs[3] == 1 && 128 * 128 * besz <= STGB && nsteps > 0);
    const bool out_img = FAST || (bp.dbias_vec != 0 && 128 * 128 * oesz <= STGB);
    float bseed[2][2][16];   // bias * log2e of this lane's row: [tile][32-key block][register]
    if (bias_img) {
        // thread `tid` fills slots tid + 256 i: 16-byte chunk c of row r <- global chunk c ^ swz(r); rows / keys past the end of the
        // (b,h) slice read back as zeros (range-checked descriptor), keys past Sk inside it are discarded by the visibility test
        const char* bbase = p.bias + ((int64_t)bb * p.bs[0] + (int64_t)hb * p.bs[1] + key0) * besz;
        const u32x4 brw = make_rsrc_words(bbase, (uint32_t)(p.bias_bytes * (besz / 2)) - (uint32_t)(key0 * besz));
        const int cpr = 128 * besz / 16;   // chunks per row: 16 or 32
        const uint32_t dst = lds_addr(ldsS) + wave * 1024;
        for (int i = 0; i < (128 * cpr) / 256; ++i) {
            const int ci = tid + i * 256, r = ci / cpr, c = (ci % cpr) ^ swz_f<128>(r);
            lds_dma16(brw, __builtin_amdgcn_readfirstlane(dst + i * 4096), (uint32_t)(((int64_t)(qblk * 128 + r) * p.bs[2]) * besz + c * 16), 0u);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
                const int e0 = t * KT + kb * 32 + 16 * hi;   // first of this lane's 16 keys inside the tile
                if (besz == 2) {
                    u32x4 w[2];
#pragma unroll
                    for (int g = 0; g < 2; ++g) w[g] = *LDS_PTR(const u32x4, ldsS + tile_off<128>(lrow, e0 / 8 + g));
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const uint32_t x = w[r >> 3][(r & 7) >> 1];
                        bseed[t][kb][r] = E::to_f32((uint16_t)((r & 1) ? (x >> 16) : (x & 0xffffu))) * kLog2e;
                    }
                } else {
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const f32x4 w = *LDS_PTR(const f32x4, ldsS + tile_off<256>(lrow, e0 / 4 + g));
#pragma unroll
                        for (int e = 0; e < 4; ++e) bseed[t][kb][4 * g + e] = w[e] * kLog2e;
                    }
                }
            }
        __syncthreads();   // the area is free for the row images
    } else if (FAST) {   // (no step to run: the seeds are never used)
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) bseed[t][kb][r] = 0.f;
    } else {   // unaligned / strided bias (or a tile that does not fit): per-element loads
        const char* brow = p.bias + (bb * p.bs[0] + hb * p.bs[1] + (int64_t)row * p.bs[2]) * besz;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float bv = 0.f;
                    const int key = key0 + t * KT + kb * 32 + 16 * hi + r;
                    if (row_ok && key < p.Sk && nsteps > 0) {
                        if (p.bias_f32) bv = reinterpret_cast<const float*>(brow)[(int64_t)key * p.bs[3]];
                        else bv = E::to_f32(reinterpret_cast<const uint16_t*>(brow)[(int64_t)key * p.bs[3]]);
                    }
                    bseed[t][kb][r] = bv * kLog2e;
                }
    }

    TileDma<D, NLD> tdK, tdV;
    tdK.init(tid, p.ks[2], true);   // key-permuted rows
    tdV.init(tid, p.vs[2], true);
    TileDma<D, NLDR> tdQ, tdD;
    if (STAGE) {
        tdQ.init(tid, p.qs[2]);
        tdD.init(tid, bp.dos[2]);
    }
    const uint32_t ldsK_w = lds_addr(ldsK) + wave * 1024, ldsV_w = lds_addr(ldsV) + wave * 1024;
    const uint32_t ldsQ_w = lds_addr(ldsQ) + wave * 1024, ldsDO_w = lds_addr(ldsDO) + wave * 1024;
    // position of a step in the walk: tile t of (b,h) number j = (b - b_lo) * nh + (h - h_lo); advanced by one step at a time (no
    // integer divisions in the loop: each costs ~40 scalar / vector instructions, and the first version did eight of them per step)
    struct Cursor {
        int s, t, j, b, h;
    };
    auto advance = [&](Cursor& c) {
        ++c.s;
        if (++c.t == nt) {
            c.t = 0;
            ++c.j;
            if (++c.h == h_lo + nh) {
                c.h = h_lo;
                ++c.b;
            }
        }
    };
    auto request = [&](const Cursor& c) {   // K / V tile of step c.s
        const int s = c.s, t = c.t, b = c.b, h = c.h;
        const char* kbase = p.k + (b * p.ks[0] + (h / p.kvg) * p.ks[1]) * 2;
        const char* vbase = p.v + (b * p.vs[0] + (h / p.kvg) * p.vs[1]) * 2;
        tdK.dma(make_rsrc_words(kbase, p.kbytes), ldsK_w + (s & 1) * TILEB, key0 + t * KT, p.ks[2]);
        tdV.dma(make_rsrc_words(vbase, p.vbytes), ldsV_w + (s & 1) * TILEB, key0 + t * KT, p.vs[2]);
    };
    auto request_rows = [&](int b, int h) {   // Q / dO rows of one (b,h) (rows past Sq read back as zeros)
        tdQ.dma(make_rsrc_words(p.q + (b * p.qs[0] + h * p.qs[1]) * 2, bp.qbytes), ldsQ_w, qblk * 128, p.qs[2]);
        tdD.dma(make_rsrc_words(bp.dout + (b * bp.dos[0] + h * bp.dos[1]) * 2, bp.dobytes), ldsDO_w, qblk * 128, bp.dos[2]);
    };
    Cursor cs{0, 0, 0, b_lo, h_lo};   // the step being computed
    Cursor cn = cs;                    // the step after it
    if (nsteps > 0) {
        request(cs);
        if (STAGE) request_rows(b_lo, h_lo);
        advance(cn);
    }

    // mask bytes of this lane's 16 keys per 32-key block of step s (no mask: all visible); requested one step ahead
    auto load_mask = [&](const Cursor& c, uint32_t (&mw)[2][4]) __attribute__((always_inline)) {
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int g = 0; g < 4; ++g) mw[kb][g] = 0x01010101u;
        if (p.mask == nullptr || c.s >= nsteps) return;
        const int t = c.t, b = c.b, h = c.h;
        const uint8_t* mrow = p.mask + (b * p.ms[0] + h * p.ms[1] + (int64_t)row * p.ms[2]);
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            const int kbase_key = key0 + t * KT + kb * 32 + 16 * hi;
            if (FAST) {   // rows 4-byte aligned; bytes past Sk / rows past Sq may be read (inside the allocation's last dword / discarded below)
                if (row_ok && kbase_key < p.Sk) {
#pragma unroll
                    for (int g = 0; g < 4; ++g)
                        if (kbase_key + 4 * g < p.Sk) mw[kb][g] = *reinterpret_cast<const uint32_t*>(mrow + kbase_key + 4 * g);
                }
            } else if (row_ok && kbase_key + 16 <= p.Sk && p.mask_vec) {
#pragma unroll
                for (int g = 0; g < 4; ++g) mw[kb][g] = *reinterpret_cast<const uint32_t*>(mrow + kbase_key + 4 * g);
            } else {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    uint32_t w = 0u;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int key = kbase_key + 4 * g + e;
                        if (row_ok && key < p.Sk && mrow[(int64_t)key * p.ms[3]] != 0) w |= 1u << (8 * e);
                    }
                    mw[kb][g] = w;
                }
            }
        }
    };
    // row statistics (and, without the row images, the Q / dO fragments) of (b,h) number j; requested one (b,h) ahead
    struct RowState {
        u32x4 q[STAGE ? 1 : KS], d[STAGE ? 1 : KS];
        float l, x;
    };
    auto load_rows = [&](int b, int h, bool in_range, RowState& rs) __attribute__((always_inline)) {
        const bool ok = row_ok && in_range;
        const char* rq = p.q + (b * p.qs[0] + h * p.qs[1] + (int64_t)row * p.qs[2]) * 2 + hi * 16;
        const char* rd = bp.dout + (b * bp.dos[0] + h * bp.dos[1] + (int64_t)row * bp.dos[2]) * 2 + hi * 16;
#pragma unroll
        for (int ks = 0; ks < (STAGE ? 0 : KS); ++ks) {
            rs.q[ks] = u32x4{0u, 0u, 0u, 0u};
            rs.d[ks] = u32x4{0u, 0u, 0u, 0u};
            if (ok) {
                rs.q[ks] = gload16(rq + ks * 32);
                rs.d[ks] = gload16(rd + ks * 32);
            }
        }
        rs.l = ok ? p.lse[(int64_t)(b * p.H + h) * p.Sq + row] : INFINITY;
        rs.x = ok ? bp.delta[(int64_t)(b * p.H + h) * p.Sq + row] : 0.f;
    };

    RowState cur, nxt;
    uint32_t mw_cur[2][4], mw_nxt[2][4];
    vec8 qf[KS], dof[KS];
    float nlse2 = 0.f, ndlt = 0.f;
    if (nsteps > 0) {
        load_rows(b_lo, h_lo, true, nxt);
        load_mask(cs, mw_nxt);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    // (b,h) after the current one (for the row prefetch)
    int nb_b = b_lo, nb_h = h_lo + 1;
    if (nb_h == h_lo + nh) {
        nb_h = h_lo;
        ++nb_b;
    }
    for (; cs.s < nsteps; advance(cs), advance(cn)) {
        const int s = cs.s, t = cs.t, j = cs.j;
        if (s + 1 < nsteps) request(cn);   // its buffer was released by the barrier that ended step s - 1
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int g = 0; g < 4; ++g) mw_cur[kb][g] = mw_nxt[kb][g];
        load_mask(cn, mw_nxt);
        const bool more_bh = j + 1 < nb * nh;
        if (t == 0) {   // a new (b,h): its rows (staged or prefetched), then the next (b,h)'s are requested
            cur = nxt;
            load_rows(nb_b, nb_h, more_bh, nxt);
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {   // Q pre-scaled by c = scale*log2e, rounded to the operand type as in every vector kernel
                u32x4 qa, da;
                if (STAGE) {
                    qa = *LDS_PTR(const u32x4, ldsQ + tile_off<D>(lrow, 2 * ks + hi));
                    da = *LDS_PTR(const u32x4, ldsDO + tile_off<D>(lrow, 2 * ks + hi));
                } else {
                    qa = cur.q[ks];
                    da = cur.d[ks];
                }
                uint16_t hq[8];
                __builtin_memcpy(hq, &qa, 16);
                f32x8 f;
#pragma unroll
                for (int e = 0; e < 8; ++e) f[e] = E::to_f32(hq[e]) * p.c;
                qf[ks] = E::cvt8(f);
                __builtin_memcpy(&dof[ks], &da, 16);
            }
            nlse2 = (cur.l == -INFINITY || cur.l == INFINITY) ? -INFINITY : -cur.l * kLog2e;   // a row without weights: P = 0
            ndlt = -cur.x;
            if (STAGE && nt == 1 && more_bh) {   // one tile per (b,h): the row images are re-requested within the step
                __syncthreads();   // every wave has its fragments
                request_rows(nb_b, nb_h);
            }
        } else if (STAGE && t == 1 && more_bh) {
            request_rows(nb_b, nb_h);   // (every wave read its fragments before the barrier that ended the t = 0 step)
        }
        if (t == nt - 1) {   // the (b,h) after the next one
            if (++nb_h == h_lo + nh) {
                nb_h = h_lo;
                ++nb_b;
            }
        }
        const int k0 = key0 + t * KT;
        const char* tK = ldsK + (s & 1) * TILEB;
        const char* tV = ldsV + (s & 1) * TILEB;
        auto tile = [&](auto T_) __attribute__((always_inline)) {
            constexpr int tt = decltype(T_)::value;
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
                const int kbase_key = k0 + kb * 32 + 16 * hi;   // register r = key kbase_key + r (key-permuted rows)
                f32x16 sacc, pacc;
#pragma unroll
                for (int r = 0; r < 16; ++r) {   // start values: S' = bias*log2e - LSE*log2e (+ q'.k), dP' = -delta (+ dO.v)
                    sacc[r] = bseed[tt][kb][r] + nlse2;
                    pacc[r] = ndlt;
                }
                const int lane_k = D == 256 ? fresh_lane_id() : lane;   // (D = 256: the 32 fragment addresses recomputed per block instead of parked in scratch)
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    const vec8 kf = lds_read_rowfrag<E, D>(tK, kb * 32 + (lane_k & 31), ks, lane_k >> 5);
                    const vec8 vf = lds_read_rowfrag<E, D>(tV, kb * 32 + (lane_k & 31), ks, lane_k >> 5);
                    sacc = E::mfma(kf, qf[ks], sacc);
                    pacc = E::mfma(vf, dof[ks], pacc);
                }
                // every element of the block visible to every lane of the wave (the usual case): no per-element test
                const bool lane_all = row_ok && kbase_key + 16 <= p.Sk && kbase_key + 15 <= vis &&
                                      (mw_cur[kb][0] & mw_cur[kb][1] & mw_cur[kb][2] & mw_cur[kb][3] & 0x01010101u) == 0x01010101u;
                if (__all(lane_all)) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) dsum[tt][kb][r] = __builtin_fmaf(fast_exp2(sacc[r]), pacc[r], dsum[tt][kb][r]);
                } else {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int key = kbase_key + r;
                        const bool show = row_ok && key < p.Sk && key <= vis && ((mw_cur[kb][r >> 2] >> (8 * (r & 3))) & 0xffu) != 0;
                        const float pv = show ? fast_exp2(sacc[r]) : 0.f;
                        dsum[tt][kb][r] = __builtin_fmaf(pv, pacc[r], dsum[tt][kb][r]);
                    }
                }
            }
        };
        if (t == 0) tile(std::integral_constant<int, 0>{});
        else tile(std::integral_constant<int, 1>{});
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the next tile (and row images, mask bytes, statistics) have landed
        __syncthreads();
    }

    // ---- store the tile: through the staging image and out in coalesced 16-byte pieces, or per lane when the rows are not aligned
    char* const obase = bp.dbias + (bb * bp.dbs[0] + hb * bp.dbs[1]) * oesz;
    if (out_img) {
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
                const int e0 = t * KT + kb * 32 + 16 * hi;
                if (oesz == 2) {
#pragma unroll
                    for (int g = 0; g < 2; ++g) {
                        f32x8 x;
#pragma unroll
                        for (int e = 0; e < 8; ++e) x[e] = dsum[t][kb][8 * g + e];
                        const vec8 y = E::cvt8(x);
                        u32x4 w;
                        __builtin_memcpy(&w, &y, 16);
                        *LDS_PTR(u32x4, ldsS + tile_off<128>(lrow, e0 / 8 + g)) = w;
                    }
                } else {
#pragma unroll
                    for (int g = 0; g < 4; ++g)
                        *LDS_PTR(f32x4, ldsS + tile_off<256>(lrow, e0 / 4 + g)) = f32x4{dsum[t][kb][4 * g], dsum[t][kb][4 * g + 1], dsum[t][kb][4 * g + 2], dsum[t][kb][4 * g + 3]};
                }
            }
        __syncthreads();
        const int cpr = 128 * oesz / 16, epc = 16 / oesz;   // chunks per row, elements per chunk
        for (int i = 0; i < (128 * cpr) / 256; ++i) {
            const int ci = tid + i * 256, r = ci / cpr, c = ci % cpr;
            const int grow = qblk * 128 + r, gkey = key0 + c * epc;
            if (grow >= p.Sq || gkey >= p.Sk) continue;
            const u32x4 w = *LDS_PTR(const u32x4, ldsS + (oesz == 4 ? tile_off<256>(r, c) : tile_off<128>(r, c)));
            char* o = obase + ((int64_t)grow * bp.dbs[2] + gkey) * oesz;
            if (gkey + epc <= p.Sk) {
                gstore16(o, w);
            } else {   // the chunk straddles Sk
                for (int e = 0; e < epc && gkey + e < p.Sk; ++e) {
                    if (oesz == 4) reinterpret_cast<uint32_t*>(o)[e] = w[e];
                    else reinterpret_cast<uint16_t*>(o)[e] = (uint16_t)(w[e >> 1] >> (16 * (e & 1)));
                }
            }
        }
        __syncthreads();   // the staging area is read out before the next tile's bias image lands in it
        continue;
    }
    if (FAST || !row_ok) continue;
    char* orow = obase + (int64_t)row * bp.dbs[2] * oesz;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            const int kfirst = key0 + t * KT + kb * 32 + 16 * hi;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                if (kfirst + r >= p.Sk) continue;
                if (dp.out_f32) {
                    reinterpret_cast<float*>(orow)[kfirst + r] = dsum[t][kb][r];
                } else {
                    const f32x4 x = {dsum[t][kb][r], 0.f, 0.f, 0.f};
                    const typename E::vec4 y = E::cvt4(x);
                    uint16_t hv[4];
                    __builtin_memcpy(hv, &y, 8);
                    reinterpret_cast<uint16_t*>(orow)[kfirst + r] = hv[0];
                }
            }
        }
    }   // tiles of this workgroup
}

}  // namespace fasn

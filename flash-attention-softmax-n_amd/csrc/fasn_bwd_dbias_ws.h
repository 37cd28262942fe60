// fasn_bwd_dbias_ws.h — gradient of a batch- / head-broadcast additive bias (see fasn_bwd_dbias.h for the mathematics and the
// tile-owner scheme), rebuilt in round 4 as a two-role pipeline for the layouts that matter (16-bit bias and gradient with
// 16-byte-movable rows, D = 64 / 128, one K/V head per query head, no mask or a mask without a row dimension = key padding):
//
//   wave A (w < 4):  S'^T = K Q'^T  seeded with bias*log2e - LSE*log2e   ->  P^T = exp2(S'^T) (hidden scores: 0)  ->  16 bit, to LDS
//   wave B (w >= 4): dP'^T = V dO^T seeded with -delta, reads P^T         ->  dsum += P^T o dP'^T   (64 fp32 registers per lane)
//
// A workgroup has 8 waves and owns a [128 rows x 128 keys] tile of one bias slice; waves w and w + 4 share a SIMD and 32 rows. It walks
// the (b,h) that read the tile - a STEP per (b,h), two 64-key UNITS per step, B one unit behind A - and the walk does not stop at the end
// of a tile: the workgroup is persistent, the step sequence runs over all of its tiles, so the K / V stream never drains (the first
// version, fasn_bwd_dbias_kernel, waited for a 32 KiB tile it had requested one 0.5 us step earlier, 8200 times per CU at config 4).
//   * K and V units come by LDS-DMA into rings of three: K a whole step (two units) ahead of wave A, V two units ahead of wave B.
//   * A lane owns a query ROW. Its Q' / dO fragments come by LDS-DMA in coalesced pieces into an 8 KiB bounce area the two waves of a
//     pair share (D = 128 then uses all 160 KiB of LDS): B asks for its step's dO rows at the start of the step's first iteration, A
//     for the next step's Q rows at the start of the second, and each moves them into its operand registers at the end of the same
//     iteration, when the MFMAs that used the previous fragments are done. The finished gradient tile leaves through the same area
//     as full 256-byte rows. (Per-lane 16-byte loads of the fragments cost the vector L1 64 accesses per instruction.)
//   * The small per-row things - LSE / delta, the tile's bias values, the mask bytes - are inline-asm loads into registers (a builtin
//     load would make hipcc put `s_waitcnt vmcnt(0)` in front of the first use and drain the K / V stream with it): issued, waited for
//     (the counted `s_waitcnt` that ends every iteration) and handed to the compiler (`retire`) inside ONE iteration, into a register
//     that is an in/out operand of the statement, so no copy of a value that has not arrived yet can be scheduled.
//   * One barrier per unit publishes P^T (two buffers) and the landed K / V units.
//   * The kernel has its own lean parameter block, and what only a new step or a new tile needs is re-read from the kernel-argument
//     segment there instead of living in scalar registers through the loop (the first version spilled 220 of them to vector lanes).
// What bounds it (config 4: 9.8 ms against 17.8, `profiles/r04_bias_gradient_*`): a step moves 128 KiB into the CU (64 KiB of rows, 64 KiB
// of K / V) for 64 MFMAs per SIMD, the CU has at most its LDS in flight, and a request takes ~3 us under this load: 37 GB/s per CU.
// Removing the MFMAs, the exponentials, the barriers or half of the requests each changes the time by less than 15 % (ablations in
// the same profile file); a variant with every request two iterations ahead needed 30 registers more than a wave has and was slower.
// Tiles no (b,h) can see (causal) are zero-filled before the walk; a unit nobody can see inside a visible tile is walked with P = 0.
// No atomics, no [B,H,L,S] buffer, deterministic.
#pragma once
#include "fasn_bwd_kernel.h"

namespace fasn {

struct DbwParams {
    const char *q, *k, *v, *dout, *bias;
    const uint8_t* mask;
    const float *lse, *delta;
    char* dbias;
    int64_t qs0, qs1, ks0, ks1, vs0, vs1, dos0, dos1, bs0, bs1, ms0, ms1, dbs0, dbs1;   // batch / head strides (elements)
    int qs2, ks2, vs2, dos2, bs2, dbs2;                                                  // row strides (elements)
    unsigned qbytes, dobytes, kbytes, vbytes, bias_bytes, mask_bytes;                    // byte extents of one (b,h) slice
    int B, H, Sq, Sk, causal;
    float c;                // scale * log2(e)
    int Bb, Hb, nqb, nkb;   // extent of the bias over batch / heads (1 = reduce over it), 128-row / 128-key blocks
    int dk, dq, dh, db;     // the grid size as digits of the tile index (kblk fastest, then qblk, hb, bb)
    int xorder;             // 1: XCD-local tile order (256 workgroups, Hb % 8 == 0, nkb % 8 == 0, nqb % 4 == 0), see next_tile
};

template <int D>
constexpr int dbias_ws_smem_bytes() {
    return 6 * KT * D * 2 + 2 * 16384 + 4 * 8192;   // K ring, V ring, two P buffers, one 8 KiB bounce area per wave pair (D = 128: all 160 KiB)
}

// asynchronous register loads (see above): OFF is the immediate byte offset; rows / keys outside the descriptor's range read as zero
template <int OFF>
FASN_DEV void aload16(u32x4& dst, u32x4 rsrc, uint32_t voff) {
    asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen offset:%3" : "+v"(dst) : "v"(voff), "s"(rsrc), "n"(OFF));
}
FASN_DEV void aload4(uint32_t& dst, u32x4 rsrc, uint32_t voff) {
    asm volatile("buffer_load_dword %0, %1, %2, 0 offen" : "+v"(dst) : "v"(voff), "s"(rsrc));
}
FASN_DEV void aload1(uint32_t& dst, u32x4 rsrc, uint32_t voff) {
    asm volatile("buffer_load_ubyte %0, %1, %2, 0 offen" : "+v"(dst) : "v"(voff), "s"(rsrc));
}

template <typename Tag, int D>
__global__ void __launch_bounds__(512, 2) fasn_bwd_dbias_ws_kernel(const DbwParams p) {
    static_assert(D == 64 || D == 128, "two-role bias gradient: head dims 64 and 128");
    using E = ET<Tag>;
    using vec8 = typename E::vec8;
    constexpr int KS = D / 16;
    constexpr int ROWB = D * 2;
    constexpr int TILEB = KT * ROWB;
    constexpr int CPR = D / 8;
    constexpr int NLD = (KT * CPR) / 512;
    constexpr int PBUF = 16384;   // one P buffer: [4 row blocks][4 x 1 KiB]

    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const ldsK = smem;                // [3][TILEB]
    char* const ldsV = smem + 3 * TILEB;    // [3][TILEB]
    char* const ldsP = smem + 6 * TILEB;    // [2][PBUF]
    char* const ldsR = smem + 6 * TILEB + 2 * PBUF;   // [4 wave pairs][8 KiB]: row images on their way to registers, gradient tiles on their way out

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31;
    const int hi = lane >> 5;
    const int role = wave >> 2;   // 0 = A, 1 = B
    const int rbw = wave & 3;     // 32-row block of this wave inside the tile's 128 rows
    // the parameter block as the kernel-argument segment holds it: what a new step / tile needs is loaded from there when it is needed
    // (the empty asm hides the pointer's origin, so the loads are not hoisted out of the loop and kept in registers)
    using CP = const __attribute__((address_space(4))) DbwParams*;
    auto rare = [&]() {
        CP kp = (CP)__builtin_amdgcn_kernarg_segment_ptr();
        asm volatile("" : "+s"(kp));
        return kp;
    };
    const bool causal = p.causal != 0;
    const int coff = p.Sk - p.Sq;
    const int nsteps = (p.Bb == 1 ? p.B : 1) * (p.Hb == 1 ? p.H : 1);
    const int ntiles = p.Bb * p.Hb * p.nqb * p.nkb;

    auto visible = [&](int qblk, int kblk) {   // some row of the tile can see some key of it
        const int last_vis = causal ? min(qblk * 128 + 127, p.Sq - 1) + coff : 0x7fffffff;
        return kblk * 128 <= last_vis;
    };

    // ---- tiles nobody can see: zeros (the gradient buffer comes uninitialised)
    if (causal) {
        for (int tile = blockIdx.x; tile < ntiles; tile += (int)gridDim.x) {
            int r = tile;
            const int kblk = r % p.nkb;
            r /= p.nkb;
            const int qblk = r % p.nqb;
            r /= p.nqb;
            const int hb = r % p.Hb, bb = r / p.Hb;
            if (visible(qblk, kblk)) continue;
            char* const obase = p.dbias + (bb * p.dbs0 + hb * p.dbs1) * 2;
            for (int i = tid; i < 128 * 16; i += 512) {
                const int grow = qblk * 128 + (i >> 4), gkey = kblk * 128 + (i & 15) * 8;
                if (grow >= p.Sq || gkey >= p.Sk) continue;
                char* o = obase + ((int64_t)grow * p.dbs2 + gkey) * 2;
                if (gkey + 8 <= p.Sk) {
                    gstore16(o, u32x4{0u, 0u, 0u, 0u});
                } else {
                    for (int e = 0; e < 8 && gkey + e < p.Sk; ++e) reinterpret_cast<uint16_t*>(o)[e] = 0;
                }
            }
        }
    }

    // ---- the walk. A step = one (b,h) of one tile; Step::fl: 1 = exists, 4 = first step of its tile, 8 = last step of its tile
    struct Step {
        int tile, j, b, h, qblk, kblk, hb, bb, fl;
    };
    auto flags_of = [&](const Step& s) { return s.tile < ntiles ? (1 | (s.j == 0 ? 4 : 0) | (s.j == nsteps - 1 ? 8 : 0)) : 0; };
    // Tile order. Workgroup i runs on XCD i % 8 (its own L2). Plain order: tiles i, i + grid, ... of the (bb, hb, qblk, kblk) index - an XCD
    // then owns an eighth of the key blocks and reads EVERY query row of every head (8 x the Q / dO bytes in L2 misses: 27 % of the L2
    // requests at config 4). XCD-local order (xorder): XCD x owns the heads x, x + 8, ...; its 32 workgroups cover 8 key blocks x 4 query
    // blocks of one head at a time and march through the query blocks first (the 8 key blocks' K / V of all batch elements stay in
    // the L2), then through the key-block groups (the head's rows come back from the memory-side cache), then to the next head.
    const int xl = (int)blockIdx.x >> 3, xcd = (int)blockIdx.x & 7;
    auto first_step = [&](Step& s) {
        if (p.xorder) {
            s.tile = 0;
            s.kblk = xl & 7;
            s.qblk = xl >> 3;
            s.hb = xcd;
            s.bb = 0;
            return;
        }
        int r = s.tile = blockIdx.x;
        s.kblk = r % p.nkb;
        r /= p.nkb;
        s.qblk = r % p.nqb;
        r /= p.nqb;
        s.hb = r % p.Hb;
        s.bb = r / p.Hb;
    };
    auto next_tile = [&](Step& s) {   // tile += gridDim.x in mixed radix (no division), then on to the next tile somebody can see
        do {
            if (p.xorder) {
                s.qblk += 4;
                if (s.qblk >= p.nqb) {
                    s.qblk = xl >> 3;
                    s.kblk += 8;
                    if (s.kblk >= p.nkb) {
                        s.kblk = xl & 7;
                        s.hb += 8;
                        if (s.hb >= p.Hb) {
                            s.hb = xcd;
                            if (++s.bb >= p.Bb) s.tile = ntiles;
                        }
                    }
                }
                continue;
            }
            s.tile += (int)gridDim.x;
            s.kblk += p.dk;
            int cy = s.kblk >= p.nkb;
            s.kblk -= cy ? p.nkb : 0;
            s.qblk += p.dq + cy;
            cy = s.qblk >= p.nqb;
            s.qblk -= cy ? p.nqb : 0;
            s.hb += p.dh + cy;
            cy = s.hb >= p.Hb;
            s.hb -= cy ? p.Hb : 0;
            s.bb += p.db + cy;
        } while (s.tile < ntiles && !visible(s.qblk, s.kblk));
    };
    auto enter_tile = [&](Step& s) {
        s.j = 0;
        s.b = p.Bb == 1 ? 0 : s.bb;
        s.h = p.Hb == 1 ? 0 : s.hb;
        s.fl = flags_of(s);
    };
    auto advance = [&](Step& s) {
        if (!(s.fl & 1)) return;
        if (++s.j < nsteps) {
            const int h_lo = p.Hb == 1 ? 0 : s.hb, h_n = p.Hb == 1 ? p.H : 1;
            if (++s.h == h_lo + h_n) {
                s.h = h_lo;
                ++s.b;
            }
            s.fl = flags_of(s);
            return;
        }
        next_tile(s);
        enter_tile(s);
    };

    // ---- K / V units straight to LDS (512 threads: NLD 16-byte chunks per thread per tensor), keys permuted inside a 32-key block so
    // that a lane's 16 accumulator registers are 16 CONSECUTIVE keys (16 hi + r)
    unsigned voffK[NLD], voffV[NLD];
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
        const int ci = tid + i * 512;
        const int r = ci / CPR, ch = (ci % CPR) ^ swz_f<D>(r);
        const int grow = (r & ~31) | (((r >> 2) & 1) << 4) | (((r >> 3) & 3) << 2) | (r & 3);
        voffK[i] = (unsigned)(grow * p.ks2 * 2 + ch * 16);
        voffV[i] = (unsigned)(grow * p.vs2 * 2 + ch * 16);
    }
    const uint32_t ldsK_w = lds_addr(ldsK) + wave * 1024, ldsV_w = lds_addr(ldsV) + wave * 1024;
    auto k_dma = [&](const Step& s, int t, int slot) {
        const u32x4 rw = make_rsrc_words(p.k + (s.b * p.ks0 + s.h * p.ks1) * 2, p.kbytes);
        const int soff = (s.kblk * 128 + t * KT) * p.ks2 * 2;
#pragma unroll
        for (int i = 0; i < NLD; ++i) lds_dma16(rw, __builtin_amdgcn_readfirstlane(ldsK_w + slot * TILEB + i * 8192), voffK[i], soff);
    };
    auto v_dma = [&](const Step& s, int t, int slot) {
        const u32x4 rw = make_rsrc_words(p.v + (s.b * p.vs0 + s.h * p.vs1) * 2, p.vbytes);
        const int soff = (s.kblk * 128 + t * KT) * p.vs2 * 2;
#pragma unroll
        for (int i = 0; i < NLD; ++i) lds_dma16(rw, __builtin_amdgcn_readfirstlane(ldsV_w + slot * TILEB + i * 8192), voffV[i], soff);
    };
    // MFMA A operand "row = key = lane & 31, 8 features of k-step s" of a unit in ring slot `slot`, key block kb: the swizzle is an XOR
    // on the chunk index, so k-step s is (address of step 0) ^ (s << 5); slot and key block only add multiples of the row size
    const uint32_t x0 = l31 * ROWB + ((hi ^ swz_f<D>(l31)) << 4);
    auto frag = [&](uint32_t base, int s, int kb) {
        const u32x4 raw = *LDS_PTR(const u32x4, (uint32_t)((base ^ (uint32_t)(s << 5)) + kb * 32 * ROWB));
        vec8 r;
        __builtin_memcpy(&r, &raw, 16);
        return r;
    };
    auto pslot = [&](int pb) { return ldsP + pb * PBUF + rbw * 4096 + lane * 16; };

    Step cur, nxt, prv;
    first_step(cur);
    if (!visible(cur.qblk, cur.kblk)) next_tile(cur);
    enter_tile(cur);
    if (!(cur.fl & 1)) return;   // (every tile of this workgroup was zero-filled above)
    nxt = cur;
    advance(nxt);
    prv = cur;
    prv.fl = 0;

    // ---- prologue: K of both units of the first step, V of its first unit
    k_dma(cur, 0, 0);
    k_dma(cur, 1, 1);
    v_dma(cur, 0, 0);

    // this wave's operand fragments (B operand: col = q row, k = 8 features): Q' for wave A, dO for wave B
    vec8 opf[KS];
    uint32_t nstat = 0u;
    float stat = 0.f;
    // Q (wave A) / dO (wave B) rows of a step: the wave's 32 rows come by LDS-DMA in coalesced 16-byte pieces into the pair's bounce area
    // (a swizzled [32][D] image, rows past Sq read back as zeros) and go on to registers at the end of the same iteration, when the
    // MFMAs that used the previous fragments are done: B asks for its step's dO in the first iteration of the step, A for the next
    // step's Q in the second. The row statistic (LSE / delta) is one asynchronous register load.
    constexpr int NR = (32 * CPR) / 64;
    char* const bounce = ldsR + rbw * 8192;
    const uint32_t bounce_a = lds_addr(bounce);
    unsigned voffR[NR];
#pragma unroll
    for (int i = 0; i < NR; ++i) {
        const int ci = lane + i * 64;
        const int r = ci / CPR, ch = (ci % CPR) ^ swz_f<D>(r);
        voffR[i] = (unsigned)(r * (role == 0 ? p.qs2 : p.dos2) * 2 + ch * 16);
    }
    auto rows_request = [&](const Step& s) {
        CP kp = rare();
        const int row0 = s.qblk * 128 + rbw * 32;
        const int64_t bh = (int64_t)s.b * kp->H + s.h;
        const int Sq = kp->Sq;
        u32x4 rw;
        int soff;
        if (role == 0) {
            rw = make_rsrc_words(kp->q + (s.b * kp->qs0 + s.h * kp->qs1) * 2, kp->qbytes);
            soff = row0 * kp->qs2 * 2;
            aload4(nstat, make_rsrc_words(kp->lse + bh * Sq, (uint32_t)Sq * 4u), (uint32_t)(row0 + l31) * 4u);
        } else {
            rw = make_rsrc_words(kp->dout + (s.b * kp->dos0 + s.h * kp->dos1) * 2, kp->dobytes);
            soff = row0 * kp->dos2 * 2;
            aload4(nstat, make_rsrc_words(kp->delta + bh * Sq, (uint32_t)Sq * 4u), (uint32_t)(row0 + l31) * 4u);
        }
#pragma unroll
        for (int i = 0; i < NR; ++i) lds_dma16(rw, __builtin_amdgcn_readfirstlane(bounce_a + i * 1024), voffR[i], soff);
    };
    int s0 = 0;   // ring slot of the current step's first unit: (2 * steps done) % 3
    auto end_iteration = [&](bool in_flight) {
        if (in_flight) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NLD) : "memory");   // this iteration's K and V units may stay in flight
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    };

    if (role == 0) {
        // =================================================== wave A ===================================================
        u32x4 braw[2][2][2];      // the tile's bias values of this lane's row, 16 bit: [unit][32-key block][8 keys], key = 16 hi + 8 g + e
        u32x4 nbias[2][2][2];     // the next tile's
        uint32_t nmk[2] = {1u, 1u};   // mask bytes of keys key0 + 64 t + lane of the next step
        uint64_t kpw[2] = {~0ull, ~0ull};
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int g = 0; g < 2; ++g) nbias[t][kb][g] = braw[t][kb][g] = u32x4{0u, 0u, 0u, 0u};
        auto a_request = [&](const Step& s) {   // everything wave A needs for a step, and for its tile if the step opens one
            rows_request(s);
            CP kp = rare();
            if (kp->mask != nullptr) {
                const u32x4 mrw = make_rsrc_words(kp->mask + (s.b * kp->ms0 + s.h * kp->ms1), kp->mask_bytes);
                aload1(nmk[0], mrw, (uint32_t)(s.kblk * 128 + lane));
                aload1(nmk[1], mrw, (uint32_t)(s.kblk * 128 + KT + lane));
            }
            if (s.fl & 4) {
                const u32x4 brw = make_rsrc_words(kp->bias + (s.bb * kp->bs0 + s.hb * kp->bs1) * 2, kp->bias_bytes);
                const int row = s.qblk * 128 + rbw * 32 + l31;
                const uint32_t vo = (uint32_t)((row * kp->bs2 + s.kblk * 128 + 16 * hi) * 2);
                aload16<0>(nbias[0][0][0], brw, vo);
                aload16<16>(nbias[0][0][1], brw, vo);
                aload16<64>(nbias[0][1][0], brw, vo);
                aload16<80>(nbias[0][1][1], brw, vo);
                aload16<128>(nbias[1][0][0], brw, vo);
                aload16<144>(nbias[1][0][1], brw, vo);
                aload16<192>(nbias[1][1][0], brw, vo);
                aload16<208>(nbias[1][1][1], brw, vo);
            }
        };
        auto a_take = [&](const Step& s) {   // behind the counted wait: what was requested becomes the operands of step s
            const int row = s.qblk * 128 + rbw * 32 + l31;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {   // Q' = Q * scale*log2e, rounded to the operand type as in every vector kernel
                const u32x4 raw = *LDS_PTR(const u32x4, bounce + tile_off<D>(l31, 2 * ks + hi));
                uint16_t hq[8];
                __builtin_memcpy(hq, &raw, 16);
                f32x8 f;
#pragma unroll
                for (int e = 0; e < 8; ++e) f[e] = E::to_f32(hq[e]) * p.c;
                opf[ks] = E::cvt8(f);
            }
            retire_loads(nstat);
            const float l = __uint_as_float(nstat);
            stat = (row >= p.Sq || l == -INFINITY || l == INFINITY) ? -INFINITY : -l * kLog2e;   // a row without weights: P = 0
            retire_loads(nmk[0]);
            retire_loads(nmk[1]);
            if (p.mask != nullptr) {
                kpw[0] = __ballot(nmk[0] != 0u);
                kpw[1] = __ballot(nmk[1] != 0u);
            }
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int g = 0; g < 2; ++g) retire_loads(nbias[t][kb][g]);
            if (s.fl & 4) {
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                        for (int g = 0; g < 2; ++g) braw[t][kb][g] = nbias[t][kb][g];
            }
        };
        auto a_unit = [&](const Step& s, auto T_, int kslot) {
            constexpr int tt = decltype(T_)::value;
            const uint32_t xk = x0 + lds_addr(ldsK) + kslot * TILEB;
            const int row0 = s.qblk * 128 + rbw * 32;
            const int k_unit0 = s.kblk * 128 + tt * KT;
            const int vis = causal ? row0 + l31 + coff : 0x7fffffff;
            const uint64_t kpb = kpw[tt];
            const bool need_mask = (causal && k_unit0 + KT - 1 > row0 + coff) || (k_unit0 + KT > p.Sk) || kpb != ~0ull;
            vec8 pf[2][2];
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
                f32x16 sacc;
#pragma unroll
                for (int r = 0; r < 16; ++r) {   // start value: bias * log2e - LSE * log2e
                    const uint32_t x = braw[tt][kb][r >> 3][(r & 7) >> 1];
                    sacc[r] = __builtin_fmaf(E::to_f32((uint16_t)((r & 1) ? (x >> 16) : (x & 0xffffu))), kLog2e, stat);
                }
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) sacc = E::mfma(frag(xk, ks, kb), opf[ks], sacc);
                const uint32_t lane_bits = (uint32_t)(kpb >> (kb * 32)) >> (16 * hi);   // this lane's 16 keys of the block
                auto elems = [&](auto MASKED) {
#pragma unroll
                    for (int t2 = 0; t2 < 2; ++t2) {
                        f32x8 x;
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            const int r = 8 * t2 + e;
                            float pv = fast_exp2(sacc[r]);
                            if (decltype(MASKED)::value) {
                                const int key = k_unit0 + kb * 32 + 16 * hi + r;
                                pv = (key < p.Sk && key <= vis && ((lane_bits >> r) & 1u)) ? pv : 0.f;
                            }
                            x[e] = pv;
                        }
                        pf[kb][t2] = E::cvt8(x);
                    }
                };
                if (need_mask) elems(std::true_type{});
                else elems(std::false_type{});
            }
            char* ps = pslot(tt);
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int t2 = 0; t2 < 2; ++t2) {
                    u32x4 w;
                    __builtin_memcpy(&w, &pf[kb][t2], 16);
                    *LDS_PTR(u32x4, ps + (kb * 2 + t2) * 1024) = w;
                }
        };

        a_request(cur);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        a_take(cur);
        __syncthreads();
        while (cur.fl & 1) {
            const int s1 = s0 == 2 ? 0 : s0 + 1, s2 = s0 == 0 ? 2 : s0 - 1;
            const bool more = (nxt.fl & 1) != 0;
            // ---- iteration (step, unit 0): A on unit 0, B on unit 1 of the previous step (the bounce area is B's)
            if (more) k_dma(nxt, 0, s2);
            v_dma(cur, 1, s1);
            a_unit(cur, std::integral_constant<int, 0>{}, s0);
            end_iteration(more);
            __syncthreads();
            // ---- iteration (step, unit 1): A on unit 1, B on unit 0; A asks for the next step's operands and takes them at the end
            if (more) {
                a_request(nxt);
                k_dma(nxt, 1, s0);
                v_dma(nxt, 0, s2);
            }
            a_unit(cur, std::integral_constant<int, 1>{}, s1);
            end_iteration(more);
            if (more) a_take(nxt);
            __syncthreads();
            cur = nxt;
            advance(nxt);
            s0 = s2;
        }
        __syncthreads();   // (wave B's last iteration)
    } else {
        // =================================================== wave B ===================================================
        f32x16 dsum[2][2];   // [unit][32-key block]: rows of this wave x keys kblk*128 + 64 t + 32 kb + 16 hi + r
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) dsum[t][kb][r] = 0.f;
        f32x16 seed;   // -delta of the lane's row in every register: start value of dP'
#pragma unroll
        for (int r = 0; r < 16; ++r) seed[r] = 0.f;
        auto b_take = [&](const Step& s) {   // behind the counted wait: the dO fragments and -delta of step s
            const int row = s.qblk * 128 + rbw * 32 + l31;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const u32x4 raw = *LDS_PTR(const u32x4, bounce + tile_off<D>(l31, 2 * ks + hi));
                __builtin_memcpy(&opf[ks], &raw, 16);
            }
            retire_loads(nstat);
            stat = row < p.Sq ? -__uint_as_float(nstat) : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) seed[r] = stat;
        };
        auto b_unit = [&](auto T_, int vslot) {
            constexpr int tt = decltype(T_)::value;
            const uint32_t xv = x0 + lds_addr(ldsV) + vslot * TILEB;
            const char* ps = pslot(tt);
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
                u32x4 pw[2];
#pragma unroll
                for (int t2 = 0; t2 < 2; ++t2) pw[t2] = *LDS_PTR(const u32x4, ps + (kb * 2 + t2) * 1024);
                f32x16 pacc = seed;
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) pacc = E::mfma(frag(xv, ks, kb), opf[ks], pacc);
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const uint32_t word = pw[r >> 3][(r & 7) >> 1];
                    const float pv = E::to_f32((uint16_t)((r & 1) ? (word >> 16) : (word & 0xffffu)));
                    dsum[tt][kb][r] = __builtin_fmaf(pv, pacc[r], dsum[tt][kb][r]);
                }
            }
        };
        // A complete tile leaves through the pair's bounce area as a [32 rows][128 keys] image in the bias's dtype, read back as full
        // 256-byte rows (16 bytes = 8 keys per lane and store), and the sums start again from zero. It is called at the END of an iteration,
        // behind the counted wait and the pick-up of this wave's dO rows: stores issued in front of the iteration's K / V requests would
        // be waited for with them (`vmcnt` retires in order), these have the whole next iteration.
        auto b_store = [&](const Step& s) {
            CP kp = rare();
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int g = 0; g < 2; ++g) {
                        f32x8 x;
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            x[e] = dsum[t][kb][8 * g + e];
                            dsum[t][kb][8 * g + e] = 0.f;
                        }
                        const vec8 y = E::cvt8(x);
                        u32x4 w;
                        __builtin_memcpy(&w, &y, 16);
                        *LDS_PTR(u32x4, bounce + tile_off<128>(l31, t * 8 + kb * 4 + 2 * hi + g)) = w;
                    }
            char* const obase = kp->dbias + (s.bb * kp->dbs0 + s.hb * kp->dbs1) * 2;
            const int dbs2 = kp->dbs2;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int ci = lane + i * 64, r = ci >> 4, c = ci & 15;
                const u32x4 w = *LDS_PTR(const u32x4, bounce + tile_off<128>(r, c));
                const int row = s.qblk * 128 + rbw * 32 + r, key = s.kblk * 128 + c * 8;
                if (row >= p.Sq || key >= p.Sk) continue;
                char* o = obase + ((int64_t)row * dbs2 + key) * 2;
                if (key + 8 <= p.Sk) {
                    gstore16(o, w);
                } else {
                    for (int e = 0; e < 8 && key + e < p.Sk; ++e) reinterpret_cast<uint16_t*>(o)[e] = (uint16_t)(w[e >> 1] >> (16 * (e & 1)));
                }
            }
        };

        __syncthreads();   // (wave A's first operands)
        while (cur.fl & 1) {
            const int s1 = s0 == 2 ? 0 : s0 + 1, s2 = s0 == 0 ? 2 : s0 - 1;
            const bool more = (nxt.fl & 1) != 0;
            // ---- iteration (step, unit 0): B finishes the previous step (its unit 1 sits in V slot (2j - 1) % 3 = s2), asks for this
            // step's dO rows and takes them at the end; a finished tile leaves behind them
            rows_request(cur);
            if (more) k_dma(nxt, 0, s2);
            v_dma(cur, 1, s1);
            if (prv.fl & 1) b_unit(std::integral_constant<int, 1>{}, s2);
            end_iteration(more);
            b_take(cur);
            if ((prv.fl & 9) == 9) b_store(prv);
            __syncthreads();
            // ---- iteration (step, unit 1): B on unit 0 of this step
            if (more) {
                k_dma(nxt, 1, s0);
                v_dma(nxt, 0, s2);
            }
            b_unit(std::integral_constant<int, 0>{}, s0);
            end_iteration(more);
            __syncthreads();
            prv = cur;
            cur = nxt;
            advance(nxt);
            s0 = s2;
        }
        b_unit(std::integral_constant<int, 1>{}, s0 == 0 ? 2 : s0 - 1);   // unit 1 of the last step
        b_store(prv);
        __syncthreads();
    }
}

}  // namespace fasn

// fasn_bwd_dq_ws.h — dQ with the three GEMMs of a score block split over TWO cooperating waves of one SIMD
// (the dQ counterpart of fasn_bwd_dkdv_ws.h; same orientation as the forward: a lane owns a query row).
//
//   wave A (w < 4):  S^T = K Q'^T (seeded with -LSE, + bias)  ->  P^T = exp2(S^T)  ->  P^T (16 bit) to LDS
//   wave B (w >= 4): dP'^T = V dO^T (seeded with -delta), reads P^T  ->  dS^T = P^T o dP'^T  ->  dQ^T += K^T dS^T
//
// A workgroup has 8 waves for 128 query rows; waves w and w + 4 share a SIMD and the same 32 rows. Wave A carries everything
// that has to do with masks and bias (B just receives zeros for hidden scores) and all the exponentials; wave B carries two of
// the three GEMMs and the only output accumulator. Neither needs more than ~200 registers, so the mask / bias modes run two
// waves per SIMD where the one-wave kernel (fasn_bwd_dq_kernel, 286 registers there) ran one.
// B runs one K/V tile behind A: the P^T of tile t is published by the barrier that ends iteration t. LDS: K in three buffers
// (tile t for A, t-1 for B's transposed reads, t+1 in flight), V in two (requested one iteration later than K: only B reads
// it), P^T in two, wave A's bias images in a private ring of two (requested two tiles ahead).
// Key-padding masks: the 64-key visibility words of the whole key range are built once per workgroup in LDS (one ballot per
// tile), which also tells where the last visible key is: trailing padded tiles are not walked at all.
#pragma once
#include "fasn_bwd_kernel.h"

namespace fasn {

constexpr int kDqWsMaxTiles = 1024;   // visibility words kept in LDS (8 KiB): Sk <= 65536 in the key-padding modes

// DROP = 1 (round 4): attention-weight dropout. Wave A draws the keep bits of its block (same hash, same (row, key quad) grouping as the
// forward: a lane owns a row, its register groups are key quads) and publishes P with the SIGN BIT SET for a dropped weight - P is never
// negative, so the bit is free - and wave B reads "kept" off the sign: dS = |P| o ((kept ? dP * 1/(1-p) : 0) - delta). B's dP accumulator
// then starts at 0 (dropout scales dP before delta is subtracted).
template <typename Tag, int D, int MODE, int DROP = 0>
__global__ void __launch_bounds__(512, 2) fasn_bwd_dq_ws_kernel(const BwdParams bp) {
    using E = ET<Tag>;
    using vec8 = typename E::vec8;
    const FwdParams& p = bp.f;
    constexpr int BM = 128;
    constexpr int TILEB = KT * D * 2;
    constexpr int KS = D / 16;
    constexpr int DB = D / 32;
    constexpr int CPR = D / 8;
    constexpr int NLD = (KT * CPR) / 512;
    static_assert(NLD >= 1, "tile too small for 512 threads");
    constexpr int PBUF = 4 * 4096;   // one P buffer: [4 row blocks][4 x 1 KiB]
    constexpr bool VBIAS = mode_has_vbias(MODE);
    constexpr bool KP = mode_has_keypad(MODE);
    constexpr bool KPERM = VBIAS;    // keys permuted inside a 32-key block so that a lane's 16 registers are 16 consecutive keys
    static_assert(MODE != MODE_GENERAL_SLOW && !mode_has_vmask(MODE), "element-load and dense-mask modes stay on the one-wave kernel");

    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const ldsK = smem;                         // [3][TILEB]
    char* const ldsV = smem + 3 * TILEB;             // [2][TILEB]
    char* const ldsP = smem + 5 * TILEB;             // [2][PBUF]
    char* const ldsB = smem + 5 * TILEB + 2 * PBUF;  // [4 A waves][2][32 rows][128 B] bias images (bias modes)
    uint64_t* const ldsKP = reinterpret_cast<uint64_t*>(ldsB + (VBIAS ? 4 * 2 * 4096 : 0));   // [ntiles] visibility words (key-padding modes)

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31;
    const int hi = lane >> 5;
    const int role = wave >> 2;   // 0 = A, 1 = B
    const int rbw = wave & 3;     // 32-row block of this wave inside the workgroup's 128 rows
#if FASN_PRIO_WS   // (round 5 A/B: static wave priority for one role of a SIMD's pair: 1 = wave B (two of the three GEMMs), 2 = wave A (exponentials, bias))
    if (role == (FASN_PRIO_WS == 1 ? 1 : 0)) __builtin_amdgcn_s_setprio(1);
#endif
    const DropSeed dsd = DROP ? drop_seed(p.seed_lo, p.seed_hi, p.rng) : DropSeed{0u, 0u};

    // Ragged key-padded batch under a batch-broadcast bias (config 4): like the forward (fasn_fwd_kernel.h, kpair_plan) a workgroup takes
    // the r-th longest and then the r-th shortest batch element, so that every workgroup of the launch walks about the same number of
    // tiles (the in-order dispatcher makes a round as long as its longest workgroup: 19 % idle CUs here). The second element runs through
    // a second inlined copy of the body - no loop-carried state (a pass loop cost 18 spilled registers in round 3).
    constexpr bool KPAIR = KP && VBIAS && !DROP;
    int bh0, qi0, bh2 = -1, kp_lead = 0;
    if (VBIAS && p.batch_inner && (p.H & 7) == 0) {   // the B workgroups that read the same bias rows run together on one XCD
        const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
        int bb = j % p.B, rest = j / p.B;
        if constexpr (KPAIR) {
            int b0 = -1, b1 = -1;
            if (kpair_plan(p, smem, tid, j % ((p.B + 1) / 2), b0, b1, kp_lead)) {
                const int np = (p.B + 1) / 2;
                if (j >= (p.H >> 3) * bp.nblk * np) return;
                rest = j / np;
                bb = b0;
                if (b1 != b0) bh2 = b1 * p.H + (rest / bp.nblk) * 8 + xcd;
            }
        }
        qi0 = rest % bp.nblk;
        bh0 = bb * p.H + (rest / bp.nblk) * 8 + xcd;
    } else {
        const bool causal0 = (MODE == MODE_CAUSAL) || (MODE >= MODE_GENERAL && p.causal);
        block_to_work_grouped(blockIdx.x, p.B * p.H, bp.nblk, (FASN_CAUSAL_GROUPS && causal0) ? causal_head_group(p.B * p.H, p.Sk, D) : 1, bh0, qi0);
    }
    auto item = [&](const int bh, const int qi, auto SECOND_) __attribute__((always_inline)) {
    constexpr bool ROT = KPAIR && decltype(SECOND_)::value;   // second element of a length pair: rotated key walk (fasn_fwd_kernel.h, kpair_plan)
    const bool causal = (MODE == MODE_CAUSAL) || (MODE >= MODE_GENERAL && p.causal);
    const int qblk = causal ? (bp.nblk - 1 - qi) : qi;
    const int b = bh / p.H, h = bh % p.H;
    const int q0 = qblk * BM;
    const int qw0 = q0 + rbw * 32;
    const int row = qw0 + l31;
    const int coff = p.Sk - p.Sq;

    const char* qbase = p.q + (b * p.qs[0] + h * p.qs[1]) * 2;
    const char* kbase = p.k + (b * p.ks[0] + (h / p.kvg) * p.ks[1]) * 2;
    const char* vbase = p.v + (b * p.vs[0] + (h / p.kvg) * p.vs[1]) * 2;
    const char* dobase = bp.dout + (b * bp.dos[0] + h * bp.dos[1]) * 2;

    int ntiles = (p.Sk + KT - 1) / KT;
    if (causal) {
        const int kmax = min(q0 + BM, p.Sq) - 1 + coff;
        ntiles = min(ntiles, kmax < 0 ? 0 : (kmax / KT + 1));
    }

    // this wave's operand fragment (B operand: col = q row, k = 8 features): Q for wave A, dO for wave B; and its row statistic
    vec8 opf[KS];
    float stat = 0.f;
    {
        const bool ok = row < p.Sq;
        const char* rp = (role == 0 ? qbase + (int64_t)row * p.qs[2] * 2 : dobase + (int64_t)row * bp.dos[2] * 2) + hi * 16;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            u32x4 a = {0u, 0u, 0u, 0u};
            if (ok) a = gload16(rp + s * 32);
            __builtin_memcpy(&opf[s], &a, 16);
        }
        if (role == 0) {
            const float l = ok ? p.lse[(int64_t)bh * p.Sq + row] : INFINITY;
            stat = (l == -INFINITY || l == INFINITY) ? -INFINITY : -l * kLog2e;   // start value of S: a row without weights gets P = 0
        } else {
            stat = ok ? -bp.delta[(int64_t)bh * p.Sq + row] : 0.f;               // start value of dP
        }
    }

    // ---- K / V tiles straight to LDS (512 threads: NLD 16-byte chunks per thread per tensor)
    unsigned voffK[NLD], voffV[NLD];
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
        const int ci = tid + i * 512;
        const int r = ci / CPR, ch = (ci % CPR) ^ swz_f<D>(r);
        const int grow = KPERM ? ((r & ~31) | (((r >> 2) & 1) << 4) | (((r >> 3) & 3) << 2) | (r & 3)) : r;
        voffK[i] = (unsigned)(grow * (int)p.ks[2] * 2 + ch * 16);
        voffV[i] = (unsigned)(grow * (int)p.vs[2] * 2 + ch * 16);
    }
    const u32x4 krw = make_rsrc_words(kbase, p.kbytes), vrw = make_rsrc_words(vbase, p.vbytes);
    const uint32_t ldsK_w = lds_addr(ldsK) + wave * 1024, ldsV_w = lds_addr(ldsV) + wave * 1024;
    auto k_dma = [&](int t, int buf) {
#pragma unroll
        for (int i = 0; i < NLD; ++i) lds_dma16(krw, __builtin_amdgcn_readfirstlane(ldsK_w + buf * TILEB + i * 8192), voffK[i], t * KT * (int)p.ks[2] * 2);
    };
    auto v_dma = [&](int t, int buf) {
#pragma unroll
        for (int i = 0; i < NLD; ++i) lds_dma16(vrw, __builtin_amdgcn_readfirstlane(ldsV_w + buf * TILEB + i * 8192), voffV[i], t * KT * (int)p.vs[2] * 2);
    };

    // ---- bias images (wave A): the wave's [32 rows][64 keys] of a tile, swizzled like a D = 64 tile, ring of two per wave
    u32x4 brw = {0u, 0u, 0u, 0u};
    unsigned bvo[4] = {0u, 0u, 0u, 0u};
    if (VBIAS) {
        brw = make_rsrc_words(p.bias + (b * p.bs[0] + h * p.bs[1]) * 2, p.bias_bytes);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int sl = i * 64 + lane, r = sl >> 3, c = (sl & 7) ^ swz_f<64>(r);
            bvo[i] = (unsigned)(((qw0 + r) * (int)p.bs[2] + c * 8) * 2);
        }
    }
    char* const img = ldsB + rbw * (2 * 4096);
    const uint32_t img_a = lds_addr(img);
    auto bias_request = [&](int t, int slot) {   // 4 vector-memory requests; tiles past the end are out of range: zeros
#pragma unroll
        for (int i = 0; i < 4; ++i) lds_dma16(brw, __builtin_amdgcn_readfirstlane(img_a + slot * 4096 + i * 1024), bvo[i], t * (KT * 2));
    };

    // ---- prologue
    if (ntiles > 0) {
        k_dma(0, 0);
        if (VBIAS && role == 0) {
            bias_request(0, 0);
            bias_request(1, 1);
        }
    }
    if (KP) kp_build_words(ldsKP, p.mask ? p.mask + (b * p.ms[0] + h * p.ms[1]) : nullptr, p.Sk, ntiles, tid, 512);   // visibility word of every 64-key tile
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
#pragma unroll
    for (int s = 0; s < KS; ++s) retire_loads(opf[s]);
    retire_loads(stat);
    if (KP) {   // trailing tiles without a visible key are not walked (every wave finds the same last tile)
        int last = -1;
        for (int t = lane; t < ntiles; t += 64)
            if (ldsKP[t] != 0ull) last = t;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) last = max(last, __shfl_xor(last, o));
        ntiles = __builtin_amdgcn_readfirstlane(last + 1);
    }
    // second element of a length pair: step t works on tile (t + rot) mod ntiles - the workgroups of a (head, query block) group, which
    // start their second elements kp_lead steps apart, then ask for the same bias tile at the same time; steps past the end map to a
    // tile behind the last key (requests out of the descriptors' range: zeros)
    int rot = 0;
    if (ROT && p.kprot && ntiles > 0) rot = (ntiles - kp_lead % ntiles) % ntiles;
    const int nt_all = (p.Sk + KT - 1) / KT;
    auto phys = [&](int t) {
        if constexpr (!ROT) return t;
        else {
            const int u = t + rot;
            return t >= ntiles ? nt_all : (u >= ntiles ? u - ntiles : u);
        }
    };
    if (ROT && role == 0 && rot != 0) {   // the prologue requested the images of tiles 0 and 1: replace them (same slots, same count)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        bias_request(phys(0), 0);
        bias_request(phys(1), 1);
    }
    if (ROT && rot != 0) {   // and K tile 0
        k_dma(phys(0), 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    if (role == 0) {   // Q' = Q * scale*log2e, rounded to the operand type (like the pre-scaled q of core/flash_attn.py:81-83)
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
    f32x16 seed;   // accumulator-shaped splat of the row statistic: the C operand of the first MFMA of every key block
#pragma unroll
    for (int r = 0; r < 16; ++r) seed[r] = stat;
    f32x16 acc[DB];   // dQ^T (wave B)
#pragma unroll
    for (int d = 0; d < DB; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[d][r] = 0.f;

    const int wave_first_vis = qw0 + coff;
    const int wave_last_vis = qw0 + 31 + coff;
    const int vis = causal ? (row + coff) : 0x7fffffff;
    const uint32_t drop_rb = DROP ? drop_row_base(dsd.lo, (uint32_t)bh, (uint32_t)row) : 0u;
    const DropThr dthr = drop_thr(DROP ? p.drop_thr : 1u);
    const uint32_t drop_rh = drop_rh_of(hi);
    // wave-uniform classification of (this wave's 32 rows) x (key tile t): identical for the A and the B wave of a row block
    auto classify = [&](int t, bool& skip, bool& need_mask, uint64_t& kp_bits) {   // t: the tile itself (phys(step))
        const int k0 = t * KT;
        skip = qw0 >= p.Sq;
        need_mask = false;
        kp_bits = ~0ull;
        if (causal) {
            skip = skip || k0 > wave_last_vis;
            need_mask = (k0 + KT - 1) > wave_first_vis;
        }
        if (k0 + KT > p.Sk) need_mask = true;
        if (KP) {
            const uint64_t w = ldsKP[t];
            kp_bits = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(w >> 32)) << 32) | (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)w);   // (the builtin returns a signed int)
            if (kp_bits == 0) skip = true;
        }
    };
    // key of register r of key block kb (inside tile t)
    auto key_of = [&](int t, int kb, int r) { return t * KT + kb * 32 + (KPERM ? 16 * hi + r : (r & 3) + 8 * (r >> 2) + 4 * hi); };
    // P exchange: (P buffer, row block) -> 4 x 1 KiB, lane * 16 bytes each
    auto pslot = [&](int pb) { return ldsP + pb * PBUF + rbw * 4096 + lane * 16; };

    // ---- wave A: key tile t
    auto tile_a = [&](const int step, auto BUF_, const int par) {
        constexpr int buf = decltype(BUF_)::value;
        const char* tK = ldsK + buf * TILEB;
        const int t = phys(step);   // the tile this step works on
        bool skip, need_mask;
        uint64_t kp_bits;
        classify(t, skip, need_mask, kp_bits);
        u32x2 braw[2][4];
        if (VBIAS) {   // this tile's image -> registers (32 bytes per key block and lane), then the slot takes tile t + 2
            const char* im = img + par * 4096;
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const u32x4 w = *LDS_PTR(const u32x4, im + tile_off<64>(l31, kb * 4 + 2 * hi + j));
                    braw[kb][2 * j] = u32x2{w[0], w[1]};
                    braw[kb][2 * j + 1] = u32x2{w[2], w[3]};
                }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the image is in registers before its slot is requested again
            bias_request(phys(step + 2), par);
        }
        if (skip) return;
        vec8 pf[2][2];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            // S' = seed (+ bias) + K Q'^T. The MFMA chain is instantiated per start-value variant, so that the common one takes the
            // seed tuple itself as the C operand of its first MFMA (no copy): a boundary tile of a key-padding mask (wave-uniform,
            // at most one per workgroup) and the bias modes build their start values per element.
            auto s_gemm = [&](f32x16 c0) {
#pragma unroll
                for (int s = 0; s < KS; ++s) {
                    const vec8 kf = lds_read_rowfrag<E, D>(tK, kb * 32 + l31, s, hi);
                    c0 = E::mfma(kf, opf[s], c0);
                }
                return c0;
            };
            f32x16 sacc;
            if (VBIAS || (KP && kp_bits != ~0ull)) {
                f32x16 c0;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float v = seed[r];
                    if (VBIAS) {
                        const uint32_t w = braw[kb][r >> 2][(r & 3) >> 1];
                        v = __builtin_fmaf(E::to_f32((uint16_t)((r & 1) ? (w >> 16) : (w & 0xffffu))), kLog2e, stat);
                    }
                    if (KP) {
                        const int bit = kb * 32 + (KPERM ? 16 * hi + r : (r & 3) + 8 * (r >> 2) + 4 * hi);
                        if (kp_bits != ~0ull) v = ((kp_bits >> bit) & 1ull) ? v : -INFINITY;   // (wave-uniform test: all-visible tiles skip the selects)
                    }
                    c0[r] = v;
                }
                sacc = s_gemm(c0);
            } else {
                sacc = s_gemm(seed);
            }
            const DropBlock<KPERM> db(drop_rb, dsd.hi, (uint32_t)((t * KT + kb * 32) >> 4), hi, drop_rh);   // DROP: the states of the lane's 16 weights of this block (fasn_common.h)
            auto elems = [&](auto MASKED) {
#pragma unroll
                for (int t2 = 0; t2 < 2; ++t2) {
                    f32x8 x;
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const int r = 8 * t2 + e;
                        float pv = fast_exp2(sacc[r]);
                        if (decltype(MASKED)::value) {
                            const int key = key_of(t, kb, r);
                            pv = ((key < p.Sk) && (key <= vis)) ? pv : 0.f;
                        }
                        if (DROP) pv = db.keep(r, dthr) ? pv : -pv;   // (sign set = dropped: what wave B reads)
                        x[e] = pv;
                    }
                    pf[kb][t2] = E::cvt8(x);
                }
            };
            if (need_mask) elems(std::true_type{});
            else elems(std::false_type{});
        }
        char* ps = pslot(par);
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int t2 = 0; t2 < 2; ++t2) {
                u32x4 w;
                __builtin_memcpy(&w, &pf[kb][t2], 16);
                *LDS_PTR(u32x4, ps + (kb * 2 + t2) * 1024) = w;
            }
    };

    // ---- wave B: key tile t (the one wave A finished in the previous iteration)
    auto tile_b = [&](const int step, auto BUF_, const int par) {
        constexpr int buf = decltype(BUF_)::value;
        const char* tK = ldsK + buf * TILEB;
        const char* tV = ldsV + par * TILEB;
        const int t = phys(step);
        bool skip, need_mask;
        uint64_t kp_bits;
        classify(t, skip, need_mask, kp_bits);
        if (skip) return;
        const char* ps = pslot(par);
        vec8 dsf[2][2];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            u32x4 pw[2];
#pragma unroll
            for (int t2 = 0; t2 < 2; ++t2) pw[t2] = *LDS_PTR(const u32x4, ps + (kb * 2 + t2) * 1024);
            f32x16 pacc;
            if (DROP) {
#pragma unroll
                for (int r = 0; r < 16; ++r) pacc[r] = 0.f;
            } else {
                pacc = seed;
            }
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                const vec8 vf = lds_read_rowfrag<E, D>(tV, kb * 32 + l31, s, hi);
                pacc = E::mfma(vf, opf[s], pacc);
            }
#pragma unroll
            for (int t2 = 0; t2 < 2; ++t2) {
                f32x8 x;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const uint32_t word = pw[t2][e >> 1];
                    const float pv = E::to_f32((uint16_t)((e & 1) ? (word >> 16) : (word & 0xffffu)));
                    if (DROP) {   // sign set = dropped weight: dS = |P| ((kept ? dP / (1-p) : 0) - delta)   (stat = -delta)
                        const float ks = __builtin_signbit(pv) ? 0.f : p.drop_scale;   // (one select: the factor)
                        x[e] = __builtin_fabsf(pv) * __builtin_fmaf(pacc[8 * t2 + e], ks, stat);
                    } else {
                        x[e] = pv * pacc[8 * t2 + e];
                    }
                }
                dsf[kb][t2] = E::cvt8(x);
                // gradient of the additive bias = dS (bias modes only): registers 8*t2 .. 8*t2+7 are 8 consecutive keys
                if (VBIAS && bp.dbias != nullptr && row < p.Sq) {
                    char* drow = bp.dbias + (b * bp.dbs[0] + h * bp.dbs[1] + (int64_t)row * bp.dbs[2]) * 2;
                    uint16_t hv[8];
                    __builtin_memcpy(hv, &dsf[kb][t2], 16);
                    const int key0 = t * KT + kb * 32 + 16 * hi + 8 * t2;
                    if (bp.dbias_vec && key0 + 8 <= p.Sk) {
                        u32x4 w;
                        __builtin_memcpy(&w, hv, 16);
                        gstore16(drow + key0 * 2, w);
                    } else {
#pragma unroll
                        for (int e = 0; e < 8; ++e)
                            if (key0 + e < p.Sk) *reinterpret_cast<uint16_t*>(drow + (key0 + e) * 2) = hv[e];
                    }
                }
            }
        }
        // dQ^T[d][q] += K^T[d][key] dS^T[key][q]
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
                for (int d = 0; d < DB; ++d) {
                    const vec8 ktf = lds_read_trfrag<E, D>(tK, kb * 32 + 16 * t2, d, lane);
                    acc[d] = E::mfma(ktf, dsf[kb][t2], acc[d]);
                }
    };

    // iteration t: A works on tile t, B on tile t-1; K tile t+1 and V tile t are in flight. Unrolled by the three K buffers so that
    // their offsets are compile-time constants; the two-deep buffers (V, P, bias images) take the parity of t at run time. Each role
    // runs its own copy of the loop (same trip count, same barriers), see fasn_bwd_dkdv_ws.h.
    using B0 = std::integral_constant<int, 0>;
    using B1 = std::integral_constant<int, 1>;
    using B2 = std::integral_constant<int, 2>;
    auto run = [&](auto ROLE_) {
        constexpr int ROLE = decltype(ROLE_)::value;
        auto body = [&](const int t, auto BA_, auto BB_, auto BN_) {
            if (t + 1 < ntiles) k_dma(phys(t + 1), decltype(BN_)::value);
            if (t < ntiles) v_dma(phys(t), t & 1);
            if (ROLE == 0) {
                if (t < ntiles) tile_a(t, BA_, t & 1);
            } else {
                if (t > 0) tile_b(t - 1, BB_, (t - 1) & 1);
            }
            // K tile t+1 and V tile t have landed. Wave A in the bias modes leaves its newest image request (4 pieces) in flight
            if (VBIAS && ROLE == 0 && t < ntiles) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        };
        for (int t = 0; t <= ntiles; t += 3) {
            body(t, B0{}, B2{}, B1{});
            if (t + 1 <= ntiles) body(t + 1, B1{}, B0{}, B2{});
            if (t + 2 <= ntiles) body(t + 2, B2{}, B1{}, B0{});
        }
    };
    if (ntiles > 0) {
        if (role == 0) run(std::integral_constant<int, 0>{});
        else run(std::integral_constant<int, 1>{});
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    // ---- epilogue: wave B writes dQ * scale
    if (role == 1 && row < p.Sq) {
        char* rp = bp.dq + (b * bp.dqs[0] + h * bp.dqs[1] + (int64_t)row * bp.dqs[2]) * 2;
#pragma unroll
        for (int d = 0; d < DB; ++d) store_block_wide<E>(rp + d * 64, acc[d], bp.scale, hi);   // 16-byte stores (round 5, fasn_common.h)
    }
    };   // item
    item(bh0, qi0, std::false_type{});
    if constexpr (KPAIR) {
        if (bh2 >= 0) {
            __syncthreads();   // every wave is done with the first element's LDS
            item(bh2, qi0, std::true_type{});
        }
    }
}

}  // namespace fasn

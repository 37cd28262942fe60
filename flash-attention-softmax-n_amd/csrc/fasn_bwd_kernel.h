// fasn_bwd_kernel.h — backward of attention-softmax_n by recomputation, deterministic (no atomics).
//
// With LSE_i = log(n + sum_j exp x_ij) saved by the forward, P_ij = exp(x_ij - LSE_i) already carries
// the "+n" of softmax_n and the softmax backward keeps its usual form
// (reference math: flash_attention_softmax_n/core/functional.py:15-29 differentiated; the reference's own
// Triton backward, flash_attn_triton.py:146-235, saves an LSE WITHOUT n — not reproduced here):
//   delta_i = dO_i . O_i
//   dV = P^T dO          dP = dO V^T          dS = P o (dP - delta)
//   dQ = scale * dS K    dK = scale * dS^T Q
//
// Three kernels:
//   fasn_bwd_delta : delta[b,h,i]                                  (replaces _bwd_preprocess, :129-143)
//   fasn_bwd_dq    : one workgroup per 32*QB*4 query rows, walks K/V tiles (same orientation as forward:
//                    a lane owns a query row; S^T, dP^T, dS^T in accumulator layout feed dQ^T += K^T dS^T)
//   fasn_bwd_dkdv  : one workgroup per 32*KB*4 keys, walks Q/dO tiles (a lane owns a key column;
//                    S, dP, dS [q][key] feed dV^T += dO^T P and dK^T += Q^T dS)
// Both big kernels reuse the forward's two LDS access patterns (row fragments by ds_read_b128 and
// transposed fragments by ds_read_b64_tr_b16 on the same swizzled tile image).
#pragma once
#include "fasn_common.h"
#include "fasn_fwd_kernel.h"

#ifndef FASN_DKDV256_FRESH
#define FASN_DKDV256_FRESH 1
#endif
namespace fasn {

struct BwdParams {
    FwdParams f;  // q,k,v,o,lse,mask,bias + strides + sizes; f.c = scale*log2e
    const char* dout;
    char* dq;
    char* dk;
    char* dv;
    float* delta;
    int64_t dos[3], dqs[3], dks[3], dvs[3];
    float scale;
    unsigned qbytes, dobytes;  // byte extent of one (b,h) Q / dO matrix (buffer descriptor range)
    int nblk;  // blocks per head of the launching kernel
    char* dbias;       // optional: dS written densely [B,H,Sq,Sk] (element type of q), key stride 1; nullptr = not wanted
    int64_t dbs[3];
    int dbias_vec;     // rows 16-byte aligned: 8 keys per store on the vector path
    float* dqacc;      // fused backward (developer library, tools/dev/fasn_bwd_fused.h): fp32 dQ accumulator [B,H,Sq,D] in the caller's workspace; nullptr = split kernels
    int skip;          // host side only (launch_bwd_one): bit 0 = dK/dV, bit 1 = dQ are launched by the caller (fasn_bwd_pipe.h kernels)
};

// ---------------------------------------------------------------------------------------------
// delta[b,h,i] = sum_d O[i][d] * dO[i][d]      (D/8 lanes per row, 16-byte loads)
template <typename Tag, int D>
__global__ void __launch_bounds__(256) fasn_bwd_delta_kernel(const BwdParams p) {
    using E = ET<Tag>;
    constexpr int LPR = D / 8;        // lanes per row
    constexpr int RPB = 256 / LPR;    // rows per block
    const int tid = threadIdx.x;
    const int sub = tid % LPR;
    const int64_t rows = (int64_t)p.f.B * p.f.H * p.f.Sq;
    const int64_t gr = (int64_t)blockIdx.x * RPB + tid / LPR;
    float acc = 0.f;
    if (gr < rows) {
        const int i = (int)(gr % p.f.Sq);
        const int bh = (int)(gr / p.f.Sq);
        const int b = bh / p.f.H, h = bh % p.f.H;
        const char* op = p.f.o + (b * p.f.os[0] + h * p.f.os[1] + (int64_t)i * p.f.os[2]) * 2 + sub * 16;
        const char* dp = p.dout + (b * p.dos[0] + h * p.dos[1] + (int64_t)i * p.dos[2]) * 2 + sub * 16;
        u32x4 a = gload16(op), d = gload16(dp);
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            acc += E::to_f32((uint16_t)(a[w] & 0xffff)) * E::to_f32((uint16_t)(d[w] & 0xffff));
            acc += E::to_f32((uint16_t)(a[w] >> 16)) * E::to_f32((uint16_t)(d[w] >> 16));
        }
    }
#pragma unroll
    for (int o = 1; o < LPR; o <<= 1) acc += __shfl_xor(acc, o);
    if (gr < rows && sub == 0) p.delta[gr] = acc;
}

// delta of a lane's row from the row's O and dO chunks, in a kernel where a lane owns a query row (round 5: the dQ kernels of head dims <= 64
// compute delta = rowsum(O o dO) themselves, in their prologue, and publish it for the dK/dV kernel that follows - fasn_bwd_delta_kernel is
// not launched there). The lane (row l31, half hi) holds the 16-byte chunks 2s + hi (s = 0 .. KS-1) of its row; the value is BIT-IDENTICAL to
// the delta kernel's: eight sequential fmas per chunk, then that kernel's xor-1 / xor-2 / ... tree over the 2 KS chunk sums.
template <typename E, int KS>
FASN_DEV float row_delta(const u32x4 (&o)[KS], const u32x4 (&d)[KS]) {
    float pr[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        float acc = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            acc += E::to_f32((uint16_t)(o[s][w] & 0xffff)) * E::to_f32((uint16_t)(d[s][w] & 0xffff));
            acc += E::to_f32((uint16_t)(o[s][w] >> 16)) * E::to_f32((uint16_t)(d[s][w] >> 16));
        }
        pr[s] = acc + __shfl_xor(acc, 32);   // chunk 2s + chunk 2s+1 (the delta kernel's xor-1 level)
    }
#pragma unroll
    for (int o2 = 1; o2 < KS; o2 <<= 1) {   // its xor-2, xor-4, ... levels
        float nx[KS];
#pragma unroll
        for (int s = 0; s < KS; ++s) nx[s] = pr[s] + pr[s ^ o2];
#pragma unroll
        for (int s = 0; s < KS; ++s) pr[s] = nx[s];
    }
    return pr[0];
}
// (D = 128 / 256: the one-wave dQ kernels are at their register limit, and so is the causal D = 32 instantiation with 64 rows per wave - 2 -> 6 spilled
// registers with the O chunks live in its prologue: delta keeps its launch there)
#ifndef FASN_DQ_FUSED_DELTA
#define FASN_DQ_FUSED_DELTA 1   // (A/B: 0 = the one-wave dQ kernels read delta from the delta kernel's launch, as before round 5)
#endif
constexpr bool dq_computes_delta(int D, int QB, int MODE) { return FASN_DQ_FUSED_DELTA && (D == 64 || D == 32); }

// shared helpers: stage a [64][D] tile (rows row0..row0+63 of one (b,h) matrix) global -> registers -> swizzled LDS image.
// Buffer loads through a per-(b,h) descriptor: fixed per-thread byte offset, tile offset in an SGPR, rows past the end of
// the matrix read back as zeros (no predication, no per-tile vector address arithmetic).
template <int D, int NLD>
struct TileStage {
    unsigned voff[NLD];
    int loff[NLD];
    // kperm: LDS row rho of a 32-row block receives global row (key) kperm(rho), see fasn_fwd_kernel.h (vector general modes)
    FASN_DEV void init(int tid, int64_t row_stride, bool kperm = false) {
        constexpr int CPR = D / 8;
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int ci = tid + i * 256;
            const int row = ci / CPR, ch = ci % CPR;
            const int grow = kperm ? ((row & ~31) | (((row >> 2) & 1) << 4) | (((row >> 3) & 3) << 2) | (row & 3)) : row;
            voff[i] = (unsigned)(grow * (int)row_stride * 2 + ch * 16);
            loff[i] = tile_off<D>(row, ch);
        }
    }
    FASN_DEV void gload(u32x4 (&st)[NLD], __amdgpu_buffer_rsrc_t rs, int row0, int64_t row_stride) const {
        const int soff = row0 * (int)row_stride * 2;
#pragma unroll
        for (int i = 0; i < NLD; ++i) st[i] = __builtin_amdgcn_raw_buffer_load_b128(rs, voff[i], soff, 0);
    }
    FASN_DEV void lstore(const u32x4 (&st)[NLD], char* tile) const {
#pragma unroll
        for (int i = 0; i < NLD; ++i) *LDS_PTR(u32x4, tile + loff[i]) = st[i];
    }
};

// The same tile image filled straight from HBM/L2 (`buffer_load_dwordx4 ... lds`, no staging registers): thread `tid`
// owns LDS slots tid + 256*i and fetches the 16-byte chunk the swizzle assigns to each. The caller waits (vmcnt) and
// barriers before the tile is read.
template <int D, int NLD>
struct TileDma {
    unsigned voff[NLD];
    FASN_DEV void init(int tid, int64_t row_stride, bool kperm = false) {
        constexpr int CPR = D / 8;
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int ci = tid + i * 256;
            const int row = ci / CPR, ch = (ci % CPR) ^ swz_f<D>(row);
            const int grow = kperm ? ((row & ~31) | (((row >> 2) & 1) << 4) | (((row >> 3) & 3) << 2) | (row & 3)) : row;
            voff[i] = (unsigned)(grow * (int)row_stride * 2 + ch * 16);
        }
    }
    FASN_DEV void dma(u32x4 rw, uint32_t tile_addr_wave, int row0, int64_t row_stride) const {
        const int soff = row0 * (int)row_stride * 2;
#pragma unroll
        for (int i = 0; i < NLD; ++i) lds_dma16(rw, __builtin_amdgcn_readfirstlane(tile_addr_wave + i * 4096), voff[i], soff);
    }
};

// ---------------------------------------------------------------------------------------------
// dQ: workgroup = 4 waves x QB x 32 query rows, loop over 64-key tiles.
#ifndef FASN_BWD_UNROLL2
#define FASN_BWD_UNROLL2 1
#endif
#ifndef FASN_DQ_SEED_D128
#define FASN_DQ_SEED_D128 2
#endif
#ifndef FASN_DQ_SEED_D32
#define FASN_DQ_SEED_D32 2
#endif
// BF32 (round 5): fp32 bias image next to 16-bit q / k / v, as in the forward (fasn_fwd_kernel.h)
template <typename Tag, int D, int QB, int MODE, int OCC, int DROP = 0, int DQ_SEED = (D >= 128 ? FASN_DQ_SEED_D128 : D == 32 ? FASN_DQ_SEED_D32 : 3), int BF32 = 0>
__global__ void __launch_bounds__(256, OCC) fasn_bwd_dq_kernel(const BwdParams bp) {
    static_assert(!BF32 || mode_has_vbias(MODE), "fp32 bias image: the vector bias modes");
    constexpr int IMGB = BF32 ? 8192 : 4096, IMGM = mode_has_vmask(MODE) ? 2048 : 0, BPC = BF32 ? 16 : 8, BW = BF32 ? 16 : 8;   // (fasn_fwd_kernel.h)
    using E = ET<Tag>;
    using vec8 = typename E::vec8;
    const FwdParams& p = bp.f;
    constexpr int BM = 4 * QB * 32;
    constexpr int TILEB = KT * D * 2;
    constexpr int KS = D / 16;
    constexpr int DB = D / 32;
    constexpr int NLD = (KT * (D / 8)) / 256;

    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const ldsK = smem;
    char* const ldsV = smem + 2 * TILEB;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31;
    const int hi = lane >> 5;
    const DropSeed dsd = DROP ? drop_seed(p.seed_lo, p.seed_hi, p.rng) : DropSeed{0u, 0u};
    const DropThr dthr = drop_thr(DROP ? p.drop_thr : 1u);
    const uint32_t drop_rh = drop_rh_of(hi);

    int bh, qi;
    const bool causal = (MODE == MODE_CAUSAL) || (MODE >= MODE_GENERAL && p.causal);
    // paired causal launch (see fasn_fwd_kernel.h): query block nblk-1-r, then block r, so that every workgroup walks the same number of tiles
    constexpr bool PAIRABLE = (MODE == MODE_CAUSAL || (FASN_VEC_PAIR && D <= 128 && mode_is_vector(MODE) && !mode_has_keypad(MODE))) && (!DROP || (FASN_DROP_PAIR && MODE == MODE_CAUSAL));   // (round 6: also the vector modes under the causal flag, and the causal dropout kernels)
    // (a causal launch that does not pair - small, or grouped K/V in the dK/dV kernel below - takes its heads in groups, blocks heaviest first across a group: fasn_common.h)
    block_to_work_grouped(blockIdx.x, p.B * p.H, (PAIRABLE && p.pair) ? (bp.nblk + 1) / 2 : bp.nblk,
                          (FASN_CAUSAL_GROUPS && causal && !(PAIRABLE && p.pair)) ? causal_head_group(p.B * p.H, p.Sk, D) : 1, bh, qi);
    const int npass = (PAIRABLE && p.pair && qi != bp.nblk - 1 - qi) ? 2 : 1;
    for (int pass = 0; pass < npass; ++pass) {
    if (pass) __syncthreads();   // the first block's last tile has been read by every wave before the buffers are refilled
    const int qblk = causal ? (pass == 0 ? bp.nblk - 1 - qi : qi) : qi;
    const int b = bh / p.H, h = bh % p.H;
    const int q0 = qblk * BM;
    const int qw0 = q0 + wave * (QB * 32);
    const int coff = p.Sk - p.Sq;

    const char* qbase = p.q + (b * p.qs[0] + h * p.qs[1]) * 2;
    const char* kbase = p.k + (b * p.ks[0] + (h / p.kvg) * p.ks[1]) * 2;
    const char* vbase = p.v + (b * p.vs[0] + (h / p.kvg) * p.vs[1]) * 2;
    const char* dobase = bp.dout + (b * bp.dos[0] + h * bp.dos[1]) * 2;
    const char* obase = p.o + (b * p.os[0] + h * p.os[1]) * 2;
    // (causal D = 32 with 64 rows per wave, at its register limit: the vector copy of this base is formed per pass, not in front of the pass
    // loop and parked in scratch across it)
    if (D == 32 && QB == 2 && MODE == MODE_CAUSAL) asm volatile("" : "+s"(obase));

    int ntiles = (p.Sk + KT - 1) / KT;
    if (causal) {
        const int kmax = min(q0 + BM, p.Sq) - 1 + coff;
        ntiles = min(ntiles, kmax < 0 ? 0 : (kmax / KT + 1));
    }

    constexpr bool FUSE_DELTA = dq_computes_delta(D, QB, MODE);   // delta = rowsum(O o dO) of the lane's rows computed here and published (row_delta above)
    vec8 qf[QB][KS], dof[QB][KS];
    u32x4 ofr[FUSE_DELTA ? QB : 1][KS];
    float lse2[QB], dlt[QB];
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
        const int row = qw0 + qb * 32 + l31;
        const bool ok = row < p.Sq;
        const char* rq = qbase + (int64_t)row * p.qs[2] * 2 + hi * 16;
        const char* rd = dobase + (int64_t)row * bp.dos[2] * 2 + hi * 16;
        const char* ro = obase + (int64_t)row * p.os[2] * 2 + hi * 16;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            u32x4 a = {0u, 0u, 0u, 0u}, d = {0u, 0u, 0u, 0u};
            if (FUSE_DELTA) ofr[qb][s] = u32x4{0u, 0u, 0u, 0u};
            if (ok) {
                a = gload16(rq + s * 32);
                d = gload16(rd + s * 32);
                if (FUSE_DELTA) ofr[qb][s] = gload16(ro + s * 32);
            }
            __builtin_memcpy(&qf[qb][s], &a, 16);
            __builtin_memcpy(&dof[qb][s], &d, 16);
        }
        float l = ok ? p.lse[(int64_t)bh * p.Sq + row] : 0.f;
        lse2[qb] = (l == -INFINITY) ? INFINITY : l * kLog2e;
        if (!FUSE_DELTA) dlt[qb] = ok ? bp.delta[(int64_t)bh * p.Sq + row] : 0.f;
    }

    f32x16 dqacc[QB][DB];
#pragma unroll
    for (int qb = 0; qb < QB; ++qb)
#pragma unroll
        for (int d = 0; d < DB; ++d)
#pragma unroll
            for (int r = 0; r < 16; ++r) dqacc[qb][d][r] = 0.f;

    u32x4 stK[NLD], stV[NLD];
    const __amdgpu_buffer_rsrc_t krs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(kbase), 0, p.kbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t vrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(vbase), 0, p.vbytes, 0x00020000);
    TileStage<D, NLD> tsK, tsV;
    tsK.init(tid, p.ks[2], mode_is_vector(MODE));
    tsV.init(tid, p.vs[2], mode_is_vector(MODE));
    // K/V tiles go straight to LDS (`buffer_load ... lds`, one tile ahead): no staging registers and no ds_write instructions.
    // At D = 128 the S / dP accumulators would not fit next to staging registers (the compiler parks them in AGPRs and pays ~130
    // v_accvgpr moves per tile); at D = 64 it is worth 2 % of the backward (same-box A/B 1.845 -> 1.78 ms with the dK/dV kernel).
    // Only the element-load kernels (MODE_GENERAL_SLOW) keep the register-staged path.
    constexpr bool DIRECT = MODE != MODE_GENERAL_SLOW;
    TileDma<D, NLD> tdK, tdV;
    const u32x4 krw = make_rsrc_words(kbase, p.kbytes), vrw = make_rsrc_words(vbase, p.vbytes);
    const uint32_t ldsK_w = lds_addr(ldsK) + wave * 1024, ldsV_w = lds_addr(ldsV) + wave * 1024;
    if (DIRECT) {
        tdK.init(tid, p.ks[2], mode_is_vector(MODE));
        tdV.init(tid, p.vs[2], mode_is_vector(MODE));
    }
    if (ntiles > 0) {
        if (DIRECT) {
            tdK.dma(krw, ldsK_w, 0, p.ks[2]);
            tdV.dma(vrw, ldsV_w, 0, p.vs[2]);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else {
            tsK.gload(stK, krs, 0, p.ks[2]);
            tsV.gload(stV, vrs, 0, p.vs[2]);
            tsK.lstore(stK, ldsK);
            tsV.lstore(stV, ldsV);
        }
    }
    __syncthreads();
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            retire_loads(qf[qb][s]);
            retire_loads(dof[qb][s]);
        }
        retire_loads(lse2[qb]);
        if constexpr (FUSE_DELTA) {
            u32x4 dch[KS];
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                retire_loads(ofr[qb][s]);
                __builtin_memcpy(&dch[s], &dof[qb][s], 16);
            }
            dlt[qb] = row_delta<E, KS>(ofr[qb], dch);
            const int row = qw0 + qb * 32 + l31;
            if (row < p.Sq && hi == 0) bp.delta[(int64_t)bh * p.Sq + row] = dlt[qb];
        } else {
            retire_loads(dlt[qb]);
        }
    }
    // Seeded accumulators (as in the forward): Q is multiplied by c = scale*log2e once, the S accumulator starts at -LSE*log2e
    // and the dP accumulator at -delta, so the element pass is p = exp2(S'), dS = p * dP' - no fma, no subtraction. The seeds are
    // accumulator-shaped splats (a lane owns one query row), constant for the whole kernel. Dropout scales dP before delta is
    // subtracted, so those instantiations only seed S; the element-load kernels (MODE_GENERAL_SLOW) keep the unseeded arithmetic.
    // Each splat costs 16 registers per query block: where the kernel is at its register limit only one (or none) is used
    // (DQ_SEED: bit 0 = S, bit 1 = dP; the vector general mode builds its S start value per element and always seeds S).
    constexpr bool VEC = mode_is_vector(MODE);      // instantiated: MODE_GENERAL (bias and/or mask images) and MODE_BIAS_KEYPAD (bias image + visibility bits)
    constexpr bool VMASK = mode_has_vmask(MODE);
    constexpr bool SEED_S = MODE != MODE_GENERAL_SLOW && (VEC || (DQ_SEED & 1));
    constexpr bool SEED_P = MODE != MODE_GENERAL_SLOW && !DROP && (DQ_SEED & 2);
    f32x16 sseed[(SEED_S && !VEC) ? QB : 1], dseed[SEED_P ? QB : 1];
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
        if (SEED_S) {
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                uint16_t hq[8];
                __builtin_memcpy(hq, &qf[qb][s], 16);
                f32x8 f;
#pragma unroll
                for (int e = 0; e < 8; ++e) f[e] = E::to_f32(hq[e]) * p.c;
                qf[qb][s] = E::cvt8(f);
            }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            if (SEED_S && !VEC) sseed[qb][r] = -lse2[qb];
            if (SEED_P) dseed[qb][r] = -dlt[qb];
        }
    }

    // MODE_GENERAL (vector path, same scheme as the forward): K/V rows staged in key-permuted order so a lane's 16
    // accumulator registers of a 32-key block are 16 consecutive keys; the wave's bias / mask image of each tile goes
    // HBM -> LDS with coalesced `buffer_load ... lds` and is read back 32 / 16 bytes per lane; the additive term
    // (bias*log2e/c, or -inf where the mask byte is clear) is the S accumulator's start value. An absent operand gets a
    // zero-range descriptor (bias reads 0) / an all-ones OR word (mask keeps everything).
    u32x4 brw, mrw;
    unsigned bvo[QB][IMGB / 1024], mvo[QB][2];
    const uint32_t nomask = (VMASK && p.mask == nullptr) ? 0x01010101u : 0u;
    const float binv = VEC ? kLog2e : 0.f;   // Q is pre-scaled: S' = bias*log2e - LSE*log2e + q'.k
    char* const ldsGB = smem + 4 * TILEB + wave * (QB * (IMGB + IMGM));
    char* const ldsGM = ldsGB + QB * IMGB;
    const uint32_t ldsGB_a = lds_addr(ldsGB), ldsGM_a = lds_addr(ldsGM);
    auto gen_dma = [&](int t) {
#pragma unroll
        for (int qb = 0; qb < QB; ++qb) {
#pragma unroll
            for (int i = 0; i < IMGB / 1024; ++i) lds_dma16(brw, __builtin_amdgcn_readfirstlane(ldsGB_a + qb * IMGB + i * 1024), bvo[qb][i], t * (KT * (BF32 ? 4 : 2)));
            if (VMASK) {
#pragma unroll
                for (int i = 0; i < 2; ++i) lds_dma16(mrw, __builtin_amdgcn_readfirstlane(ldsGM_a + qb * 2048 + i * 1024), mvo[qb][i], t * KT);
            }
        }
    };
    if (VEC) {
        const char* bb = p.bias ? p.bias + (b * p.bs[0] + h * p.bs[1]) * (BF32 ? 4 : 2) : p.q;
        const char* mb = (VMASK && p.mask) ? reinterpret_cast<const char*>(p.mask) + (b * p.ms[0] + h * p.ms[1]) : p.q;
        brw = make_rsrc_words(bb, p.bias ? p.bias_bytes : 0u);
        mrw = make_rsrc_words(mb, (VMASK && p.mask) ? p.mask_bytes : 0u);
#pragma unroll
        for (int qb = 0; qb < QB; ++qb) {
#pragma unroll
            for (int i = 0; i < IMGB / 1024; ++i) {
                const int sl = i * 64 + lane, row = sl / BPC, c = (sl % BPC) ^ (BF32 ? swz_f<128>(row) : swz_f<64>(row));
                bvo[qb][i] = (unsigned)((qw0 + qb * 32 + row) * (int)p.bs[2] * (BF32 ? 4 : 2) + c * 16);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int sl = i * 64 + lane, row = sl >> 2, c = (sl & 3) ^ swz_f<32>(row);
                mvo[qb][i] = (unsigned)((qw0 + qb * 32 + row) * (int)p.ms[2] + c * 16);
            }
        }
        gen_dma(0);
    }

    const int wave_first_vis = qw0 + coff;
    const int wave_last_vis = qw0 + QB * 32 - 1 + coff;
    uint32_t drop_rb[QB];   // DROP: the row part of the hash input, once per row block (two 32-bit multiplies)
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) drop_rb[qb] = DROP ? drop_row_base(dsd.lo, (uint32_t)bh, (uint32_t)(qw0 + qb * 32 + l31)) : 0u;

    // MODE_KEYPAD (mask = one byte per key of the (b,h), no bias), as in the forward: each lane fetches the byte of key
    // k0 + lane one tile ahead, a ballot makes the tile's visibility word; all-visible tiles are plain, all-hidden ones skipped
    constexpr bool KP = mode_has_keypad(MODE);
    __amdgpu_buffer_rsrc_t kprs;
    uint32_t kp_next = 0;
    const uint32_t kp_or = (KP && p.mask == nullptr) ? 1u : 0u;   // bias-only call through the bias + key-padding instantiation: every key visible
    if (KP) {
        kprs = __builtin_amdgcn_make_buffer_rsrc(p.mask ? const_cast<uint8_t*>(p.mask + (b * p.ms[0] + h * p.ms[1])) : reinterpret_cast<uint8_t*>(const_cast<char*>(p.q)),
                                                 0, p.mask ? (unsigned)p.Sk : 0u, 0x00020000);
        kp_next = (uint32_t)(uint8_t)__builtin_amdgcn_raw_buffer_load_b8(kprs, lane, 0, 0) | kp_or;
    }

    // the loop is unrolled by its two LDS buffers: the buffer offset is a compile-time constant and folds into the ds_read
    // immediates instead of costing VALU adds per LDS address (FASN_BWD_UNROLL2 = 0: dynamic buffer index, for A/B)
    using Buf0 = std::integral_constant<int, 0>;
    using Buf1 = std::integral_constant<int, FASN_BWD_UNROLL2 ? 1 : 0>;
    auto ktile_body = [&](const int t, auto BUF_) {
        const int buf = FASN_BWD_UNROLL2 ? decltype(BUF_)::value : (t & 1);
        const int k0 = t * KT;
        uint64_t kp_bits = ~0ull;
        if (KP && !VEC) {
            kp_bits = __ballot(kp_next != 0);
            kp_next = (uint32_t)(uint8_t)__builtin_amdgcn_raw_buffer_load_b8(kprs, lane, (t + 1) * KT, 0) | kp_or;
        }
        uint32_t mraw[QB][2][4];
        uint32_t braw[QB][2][BW];   // the lane's 16 keys of a 32-key block: 8 dwords of 16-bit pairs, or 16 fp32 values
        if (VEC) {   // unconditional, also for skipped tiles: the request / wait pattern is the same for every tile
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this tile's image (and its visibility bytes) have landed
            if (KP) {   // the next tile's bytes are requested after the wait, so it does not cover their latency
                kp_bits = __ballot(kp_next != 0);
                kp_next = (uint32_t)(uint8_t)__builtin_amdgcn_raw_buffer_load_b8(kprs, lane, (t + 1) * KT, 0) | kp_or;
            }
#pragma unroll
            for (int qb = 0; qb < QB; ++qb)
#pragma unroll
                for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
                    for (int j = 0; j < BW / 4; ++j) {
                        const u32x4 w = *LDS_PTR(const u32x4, ldsGB + qb * IMGB + (BF32 ? tile_off<128>(l31, kb * 8 + 4 * hi + j) : tile_off<64>(l31, kb * 4 + 2 * hi + j)));
#pragma unroll
                        for (int e = 0; e < 4; ++e) braw[qb][kb][4 * j + e] = w[e];
                    }
                    if (VMASK) {
                        const u32x4 w = *LDS_PTR(const u32x4, ldsGM + qb * 2048 + tile_off<32>(l31, kb * 2 + hi));
#pragma unroll
                        for (int g = 0; g < 4; ++g) mraw[qb][kb][g] = w[g] | nomask;
                    }
                }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            gen_dma(t + 1);                          // past-the-end tiles are out of range: zeros
            if (DIRECT) {
                tdK.dma(krw, ldsK_w + (buf ^ 1) * TILEB, k0 + KT, p.ks[2]);
                tdV.dma(vrw, ldsV_w + (buf ^ 1) * TILEB, k0 + KT, p.vs[2]);
            } else {
                tsK.gload(stK, krs, k0 + KT, p.ks[2]);   // past-the-end tiles read back as zeros
                tsV.gload(stV, vrs, k0 + KT, p.vs[2]);
            }
        } else if (DIRECT) {   // buffer buf^1 was released by the barrier that ended the previous tile
            tdK.dma(krw, ldsK_w + (buf ^ 1) * TILEB, k0 + KT, p.ks[2]);
            tdV.dma(vrw, ldsV_w + (buf ^ 1) * TILEB, k0 + KT, p.vs[2]);
        } else if (t + 1 < ntiles) {
            tsK.gload(stK, krs, k0 + KT, p.ks[2]);
            tsV.gload(stV, vrs, k0 + KT, p.vs[2]);
        }
        bool skip = false, need_mask = false;
        if (causal) {
            skip = k0 > wave_last_vis;
            need_mask = (k0 + KT - 1) > wave_first_vis;
        }
        if (k0 + KT > p.Sk) need_mask = true;
        if (MODE == MODE_GENERAL_SLOW) need_mask = true;
        if (KP && kp_bits == 0) skip = true;

        if (!skip) {
            const char* tK = ldsK + buf * TILEB;
            const char* tV = ldsV + buf * TILEB;
            // one 32-key block at a time: S and dP (2*KS MFMAs per query block), then their element pass, so only ONE block's
            // accumulators (2 x 16 registers per query block) are live at a time and block 1's MFMAs run under block 0's arithmetic
            vec8 dsf[QB][2][2];
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
                f32x16 sacc[QB], pacc[QB];
#pragma unroll
                for (int qb = 0; qb < QB; ++qb)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        if (VEC) {
                            const uint32_t w = braw[qb][kb][BF32 ? r : (r >> 1)];
                            const float v = __builtin_fmaf(BF32 ? __uint_as_float(w) : E::to_f32((uint16_t)((r & 1) ? (w >> 16) : (w & 0xffffu))), binv, -lse2[qb]);
                            if (VMASK) sacc[qb][r] = ((mraw[qb][kb][r >> 2] >> (8 * (r & 3))) & 0xffu) ? v : -INFINITY;
                            else if (KP) sacc[qb][r] = (((uint32_t)(kp_bits >> (32 * kb + 16 * hi)) >> r) & 1u) ? v : -INFINITY;   // key-permuted rows: register r = key 16*hi + r
                            else sacc[qb][r] = v;
                        } else if (KP) {   // bit (r&3) + 8(r>>2) + 4hi of this 32-key block
                            const uint32_t w = (uint32_t)(kp_bits >> (32 * kb)) >> (4 * hi);
                            sacc[qb][r] = ((w >> ((r & 3) + 8 * (r >> 2))) & 1u) ? (SEED_S ? sseed[qb][r] : 0.f) : -INFINITY;
                        } else {
                            sacc[qb][r] = SEED_S ? sseed[qb][r] : 0.f;
                        }
                        pacc[qb][r] = SEED_P ? dseed[qb][r] : 0.f;
                    }
#pragma unroll
                for (int s = 0; s < KS; ++s) {
                    vec8 kf = lds_read_rowfrag<E, D>(tK, kb * 32 + l31, s, hi);
                    vec8 vf = lds_read_rowfrag<E, D>(tV, kb * 32 + l31, s, hi);
#pragma unroll
                    for (int qb = 0; qb < QB; ++qb) {
                        sacc[qb] = E::mfma(kf, qf[qb][s], sacc[qb]);
                        pacc[qb] = E::mfma(vf, dof[qb][s], pacc[qb]);
                    }
                }
#pragma unroll
                for (int qb = 0; qb < QB; ++qb) {
                    const int row = qw0 + qb * 32 + l31;
                    const int vis = causal ? (row + coff) : 0x7fffffff;
                    // DROP: the (row, key group) states of the lane's 16 weights of this block (fasn_common.h: DropBlock)
                    const DropBlock<VEC> db(drop_rb[qb], dsd.hi, (uint32_t)((k0 + kb * 32) >> 4), hi, drop_rh);
                    auto elems = [&](auto MASKED) {
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            float y = sacc[qb][r] * p.c;
                            bool show = true;
                            if (decltype(MASKED)::value) {
                                const int key = k0 + kb * 32 + (VEC ? 16 * hi + r : (r & 3) + 8 * (r >> 2) + 4 * hi);
                                show = (key < p.Sk) && (key <= vis);
                                if (MODE == MODE_GENERAL_SLOW) {
                                    const bool inb = show && (row < p.Sq);
                                    if (p.bias != nullptr && inb) {
                                        const int64_t bo = b * p.bs[0] + h * p.bs[1] + (int64_t)row * p.bs[2] + (int64_t)key * p.bs[3];
                                        float bv;
                                        if (p.bias_f32) bv = reinterpret_cast<const float*>(p.bias)[bo];
                                        else bv = E::to_f32(reinterpret_cast<const uint16_t*>(p.bias)[bo]);
                                        y = __builtin_fmaf(bv, kLog2e, y);
                                    }
                                    if (p.mask != nullptr && inb) {
                                        const int64_t mo = b * p.ms[0] + h * p.ms[1] + (int64_t)row * p.ms[2] + (int64_t)key * p.ms[3];
                                        show = p.mask[mo] != 0;
                                    }
                                }
                            }
                            float pv = (MODE == MODE_GENERAL_SLOW) ? fast_exp2(y - lse2[qb])
                                       : SEED_S ? fast_exp2(sacc[qb][r]) : fast_exp2(__builtin_fmaf(sacc[qb][r], p.c, -lse2[qb]));
                            if (decltype(MASKED)::value) pv = show ? pv : 0.f;
                            float dp = pacc[qb][r];
                            if (DROP) {   // same keep bits as the forward (same lane layout: lane = row)
                                dp = db.keep(r, dthr) ? dp * p.drop_scale : 0.f;
                            }
                            sacc[qb][r] = SEED_P ? pv * dp : pv * (dp - dlt[qb]);
                        }
                    };
                    // (dropout instantiations: ONE copy of the element pass - the masked one; two copies keep two sets of hash temporaries live)
                    if (DROP || need_mask) elems(std::true_type{});
                    else elems(std::false_type{});
#pragma unroll
                    for (int t2 = 0; t2 < 2; ++t2) {
                        f32x8 x;
#pragma unroll
                        for (int e = 0; e < 8; ++e) x[e] = sacc[qb][8 * t2 + e];
                        dsf[qb][kb][t2] = E::cvt8(x);
                        // gradient of the additive bias = dS (only the mask / bias instantiations carry this code)
                        if ((VEC || MODE == MODE_GENERAL_SLOW) && bp.dbias != nullptr && row < p.Sq) {
                            char* drow = bp.dbias + (b * bp.dbs[0] + h * bp.dbs[1] + (int64_t)row * bp.dbs[2]) * 2;
                            uint16_t hv[8];
                            __builtin_memcpy(hv, &dsf[qb][kb][t2], 16);
                            if (VEC) {   // registers 8*t2 .. 8*t2+7 = 8 consecutive keys
                                const int key0 = k0 + kb * 32 + 16 * hi + 8 * t2;
                                if (bp.dbias_vec && key0 + 8 <= p.Sk) {
                                    u32x4 w;
                                    __builtin_memcpy(&w, hv, 16);
                                    gstore16(drow + key0 * 2, w);
                                } else {
#pragma unroll
                                    for (int e = 0; e < 8; ++e)
                                        if (key0 + e < p.Sk) *reinterpret_cast<uint16_t*>(drow + (key0 + e) * 2) = hv[e];
                                }
                            } else {
#pragma unroll
                                for (int e = 0; e < 8; ++e) {
                                    const int r = 8 * t2 + e, key = k0 + kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                                    if (key < p.Sk) *reinterpret_cast<uint16_t*>(drow + key * 2) = hv[e];
                                }
                            }
                        }
                    }
                }
            }
            // dQ^T[d][q] += K^T[d][key] dS^T[key][q]
            // (D = 256 element-load mode: the addresses of the transposed reads from a fresh lane id, see the dK/dV kernel)
            const int lane_t = (D == 256 && MODE == MODE_GENERAL_SLOW && FASN_DKDV256_FRESH) ? fresh_lane_id() : lane;
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
                    for (int d = 0; d < DB; ++d) {
                        vec8 ktf = lds_read_trfrag<E, D>(tK, kb * 32 + 16 * t2, d, lane_t);
#pragma unroll
                        for (int qb = 0; qb < QB; ++qb) dqacc[qb][d] = E::mfma(ktf, dsf[qb][kb][t2], dqacc[qb][d]);
                    }
        }
        if (DIRECT) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the next K / V tiles have landed
        } else if (VEC || t + 1 < ntiles) {
            tsK.lstore(stK, ldsK + (buf ^ 1) * TILEB);
            tsV.lstore(stV, ldsV + (buf ^ 1) * TILEB);
        }
        __syncthreads();
    };
    if constexpr (FASN_BWD_UNROLL2 != 0) {
        for (int t = 0; t < ntiles; t += 2) {
            ktile_body(t, Buf0{});
            if (t + 1 < ntiles) ktile_body(t + 1, Buf1{});
        }
    } else {
        for (int t = 0; t < ntiles; ++t) ktile_body(t, Buf0{});
    }

    if (VEC) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the image requested for the tile past the end has landed
    char* dqbase = bp.dq + (b * bp.dqs[0] + h * bp.dqs[1]) * 2;
    // D = 32 with 64 rows per wave sits at its 256 registers: the output row addresses derive from a FRESH lane id (fasn_common.h) - computed
    // from the lane id of the kernel entry they are loop invariant, the compiler forms them in front of the tile loop and parks them in
    // scratch across it (round 5: 7 / 2 spilled registers in the plain / causal instantiation, the reference's own Triton test shape)
    const int el = (D == 32 && QB == 2) ? fresh_lane_id() : lane;
    const int e31 = el & 31, ehi = el >> 5;
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
        const int row = qw0 + qb * 32 + e31;
        if (row < p.Sq) {
            char* rp = dqbase + (int64_t)row * bp.dqs[2] * 2;
#pragma unroll
            for (int d = 0; d < DB; ++d) store_block_narrow<E>(rp + d * 64, dqacc[qb][d], bp.scale, ehi);   // (8-byte stores: several instantiations at their register limit, fasn_common.h)
        }
    }
    }   // pass
}

// ---------------------------------------------------------------------------------------------
// dK, dV: workgroup = 4 waves x KB x 32 keys, loop over 64-row Q/dO tiles.
constexpr int QT = 64;  // query rows per tile

// GQA = 1 (grouped-query attention, kvg query heads per K/V head): one workgroup per (batch, K/V head, key block) walks the q-tiles
// of all query heads of its group, one head after the other, into the same fp32 accumulators: dK / dV come out per K/V head.
// DH = 2 (D = 256): two workgroups per key block, each with the full S / dP but HALF of the features of dK and dV (2 x 64 instead of
// 2 x 128 accumulator registers); the grid is doubled, block 2j + v owns feature half v.
// BF32 (round 5): fp32 bias next to 16-bit q / k / v on the vector path - the workgroup's additive tile is [64 rows][BN keys] fp32, row major
// (bias, or -inf where the mask byte is clear), moved in 16-byte pieces (4 keys), and a lane reads the 16 rows of ITS key column with plain
// ds_read_b32 (consecutive lanes, consecutive dwords: no bank conflict) instead of the 16-bit transposing read
template <typename Tag, int D, int KB, int MODE, int OCC, int DROP = 0, int GQA = 0, int DH = 1, int BF32 = 0>
__global__ void __launch_bounds__(256, OCC) fasn_bwd_dkdv_kernel(const BwdParams bp) {
    static_assert(!BF32 || mode_has_vbias(MODE), "fp32 additive tile: the vector bias modes");
    using E = ET<Tag>;
    using vec8 = typename E::vec8;
    const FwdParams& p = bp.f;
    constexpr int BN = 4 * KB * 32;
    constexpr int TILEB = QT * D * 2;
    constexpr int KS = D / 16;
    constexpr int DB = D / 32 / DH;   // feature blocks of this workgroup
    constexpr int NLD = (QT * (D / 8)) / 256;

    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const ldsQ = smem;                    // [2][TILEB]
    char* const ldsDO = smem + 2 * TILEB;       // [2][TILEB]
    float* const ldsLse = reinterpret_cast<float*>(smem + 4 * TILEB);  // [2][QT] lse*log2e
    float* const ldsDlt = ldsLse + 2 * QT;                             // [2][QT]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31;
    const int hi = lane >> 5;
    const DropSeed dsd = DROP ? drop_seed(p.seed_lo, p.seed_hi, p.rng) : DropSeed{0u, 0u};
    const DropThr dthr = drop_thr(DROP ? p.drop_thr : 1u);
    const DropLane dlane = drop_lane((int)(threadIdx.x & 31));   // (a wave's key blocks start at multiples of 32: the key's place in its 16-key group is lane & 15)

    const int kvg = GQA ? p.kvg : 1;
    const int Hkv = p.H / kvg;
    int bhk, kblk0;
    const int d0 = DH > 1 ? (int)(blockIdx.x % DH) * DB : 0;   // first feature block of this workgroup
    // paired causal launch (see fasn_fwd_kernel.h): key block r (seen by the most query rows), then block nblk-1-r
    constexpr bool PAIRABLE = (MODE == MODE_CAUSAL || (FASN_VEC_PAIR && mode_is_vector(MODE) && !mode_has_keypad(MODE))) && (!DROP || (FASN_DROP_PAIR && MODE == MODE_CAUSAL)) && !GQA && DH == 1;
    const bool causal_l = (MODE == MODE_CAUSAL) || (MODE >= MODE_GENERAL && p.causal);
    block_to_work_grouped(DH > 1 ? (int)(blockIdx.x / DH) : (int)blockIdx.x, p.B * Hkv, (PAIRABLE && p.pair) ? (bp.nblk + 1) / 2 : bp.nblk,
                          (FASN_CAUSAL_GROUPS && causal_l && !(PAIRABLE && p.pair)) ? causal_head_group(p.B * Hkv, p.Sq * kvg, D) : 1, bhk, kblk0);
    const int npass = (PAIRABLE && p.pair && kblk0 != bp.nblk - 1 - kblk0) ? 2 : 1;
    for (int pass = 0; pass < npass; ++pass) {
    if (pass) __syncthreads();   // the first block's last tile has been read by every wave before the buffers are refilled
    const int kblk = pass == 0 ? kblk0 : bp.nblk - 1 - kblk0;
    const bool causal = (MODE == MODE_CAUSAL) || (MODE >= MODE_GENERAL && p.causal);
    const int b = bhk / Hkv, hk = bhk % Hkv;
    const int kw0 = kblk * BN + wave * (KB * 32);  // first key of this wave
    const int coff = p.Sk - p.Sq;

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

    for (int g = 0; g < kvg; ++g) {   // the query heads of this K/V head (one trip without grouping)
    const int h = hk * kvg + g, bh = b * p.H + h;
    const char* qbase = p.q + (b * p.qs[0] + h * p.qs[1]) * 2;
    const char* kbase = p.k + (b * p.ks[0] + (h / p.kvg) * p.ks[1]) * 2;
    const char* vbase = p.v + (b * p.vs[0] + (h / p.kvg) * p.vs[1]) * 2;
    const char* dobase = bp.dout + (b * bp.dos[0] + h * bp.dos[1]) * 2;
    const float* lsebase = p.lse + (int64_t)bh * p.Sq;
    const float* dltbase = bp.delta + (int64_t)bh * p.Sq;
    if (g > 0) __syncthreads();   // the previous head's last tile has been read by every wave before its buffers are refilled

    // query tiles that can see this key block: rows i with i + coff >= first key
    int ntq = (p.Sq + QT - 1) / QT;
    int tq0 = 0;
    if (causal) {
        const int first_row = kblk * BN - coff;
        tq0 = first_row <= 0 ? 0 : first_row / QT;
    }
    if (mode_has_keypad(MODE)) {   // a workgroup none of whose keys is visible (the padded tail of a batch element) walks no q-tile at all
        bool any = false;
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) {
            const int key = kw0 + kb * 32 + l31;
            any = any || (key < p.Sk && (p.mask == nullptr || p.mask[b * p.ms[0] + h * p.ms[1] + key] != 0));
        }
        if (__syncthreads_or(any ? 1 : 0) == 0) ntq = tq0;
    }

    // K / V fragments of this wave's keys (B operand: col = key = lane&31, k = 8 contiguous features)
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

    // Seeded accumulators: K (held in registers, used for S only) is multiplied by c = scale*log2e once; the S accumulator of a
    // query-row block starts at -LSE*log2e and the dP accumulator at -delta of the register's row (the 16 per-row values a lane
    // reads from LDS anyway), so the element pass is p = exp2(S'), dS = p * dP'. Dropout scales dP first: those kernels seed S only.
    constexpr bool SEED_S = MODE != MODE_GENERAL_SLOW;
    constexpr bool SEED_P = SEED_S && !DROP;
    u32x4 stQ[NLD], stD[NLD];
    const __amdgpu_buffer_rsrc_t qrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(qbase), 0, bp.qbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t drs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(dobase), 0, bp.dobytes, 0x00020000);
    TileStage<D, NLD> tsQ, tsD;
    tsQ.init(tid, p.qs[2]);
    tsD.init(tid, bp.dos[2]);
    // Q / dO tiles go straight to LDS (the vector general mode needs the staging registers for its additive tile; the plain
    // modes save the ds_write instructions); the element-load kernels keep the register-staged path
    constexpr bool DIRECT = MODE != MODE_GENERAL_SLOW;
    TileDma<D, NLD> tdQ, tdD;
    const u32x4 qrw = make_rsrc_words(qbase, bp.qbytes), drw = make_rsrc_words(dobase, bp.dobytes);
    const uint32_t ldsQ_w = lds_addr(smem) + wave * 1024, ldsDO_w = ldsQ_w + 2 * TILEB;
    if (DIRECT) {
        tdQ.init(tid, p.qs[2]);
        tdD.init(tid, bp.dos[2]);
    }
    float stL = 0.f, stX = 0.f;
    auto stats_gload = [&](int row0) {
        if (tid < QT) {
            // (grouped K/V at D = 256 or with dropout: the row offset from a fresh lane id - kept across the head loop it was the one value that went to scratch)
            const int gr = row0 + ((GQA && (D == 256 || DROP)) ? wave * 64 + fresh_lane_id() : tid);
            float l = 0.f, x = 0.f;
            if (gr < p.Sq) {
                l = lsebase[gr];
                x = dltbase[gr];
            }
            // seeded kernels keep the NEGATED statistics in LDS: they are the start values of the S / dP accumulators
            stL = (l == -INFINITY) ? INFINITY : l * kLog2e;
            if (SEED_S) stL = -stL;
            stX = SEED_P ? -x : x;
        }
    };
    auto stats_lstore = [&](int buf) {
        if (tid < QT) {
            ldsLse[buf * QT + tid] = stL;
            ldsDlt[buf * QT + tid] = stX;
        }
    };

    if (tq0 < ntq) {
        if (DIRECT) {
            tdQ.dma(qrw, ldsQ_w, tq0 * QT, p.qs[2]);
            tdD.dma(drw, ldsDO_w, tq0 * QT, bp.dos[2]);
        } else {
            tsQ.gload(stQ, qrs, tq0 * QT, p.qs[2]);
            tsD.gload(stD, drs, tq0 * QT, bp.dos[2]);
        }
        stats_gload(tq0 * QT);
        if (!DIRECT) {
            tsQ.lstore(stQ, ldsQ);
            tsD.lstore(stD, ldsDO);
        }
        stats_lstore(0);
    }

    // MODE_GENERAL (vector path): a lane owns a key column and register r a query row - the transpose of how bias / mask
    // rows lie in memory. Per q-tile the workgroup stages ONE combined additive tile [64 rows][BN keys] of 16-bit values in
    // LDS (bias where the mask byte is set, -inf where it is clear; coalesced 16-byte bias / 8-byte mask buffer loads,
    // rows or keys past the end read back as "hidden") and every wave fetches its 32 columns with the same transposed
    // read that feeds the MFMAs (ds_read_b64_tr_b16: 4 consecutive rows of one key column = accumulator registers 4g..4g+3).
    // The tile initialises the S accumulator (S' = add*log2e/c + q.k), so the element pass is the plain one.
    constexpr bool VEC = mode_is_vector(MODE);      // MODE_GENERAL, MODE_BIAS_KEYPAD (the additive tile is the bias alone; the mask is a per-lane flag)
    constexpr bool VMASK = mode_has_vmask(MODE);
    constexpr int BN_ = 4 * KB * 32;
    constexpr int KPC = BF32 ? 4 : 8;               // keys per 16-byte chunk of the bias
    constexpr int ADDB = QT * BN_ * (BF32 ? 4 : 2); // bytes of one additive tile = BN_/128 swizzled [64][128] 16-bit images, or [64][BN_] fp32 row major
    constexpr int ACH = (QT * BN_ / KPC) / 256;     // chunks per thread
    char* const ldsAdd = smem + 4 * TILEB + 4 * QT * 4;   // [ADD_BUFS][ADDB]
    // D = 256: ONE additive tile (four 32 KiB Q / dO buffers + two 16 KiB tiles would be 161 KiB); the next tile then waits in its staging
    // registers for a barrier of its own behind the q tile's last read. Elsewhere two, filled ahead of the barrier that ends the q tile.
    constexpr int ADD_BUFS = D == 256 ? 1 : 2;
    __amdgpu_buffer_rsrc_t brs, mrs;
    unsigned abvo0 = 0, amvo0 = 0;   // chunk 0 of this thread; chunk i is RSTEP rows further down (wave-uniform offset)
    constexpr int RSTEP = 256 / (BN_ / KPC);
    const int arow0 = tid / (BN_ / KPC), akc = tid % (BN_ / KPC);
    u32x4 stA[ACH];
    u32x2 stM[ACH];   // mask bytes of the chunk's keys (fp32: the low word only)
    const uint32_t nomask = (!VMASK || p.mask == nullptr) ? 0x01010101u : 0u;
    const float binv = VEC ? kLog2e : 0.f;   // K is pre-scaled: S' = add*log2e - LSE*log2e + q.k'
    const uint32_t ninf16 = std::is_same<Tag, bf16_tag>::value ? 0xFF80u : 0xFC00u;   // -inf in the 16-bit type
    if (VEC) {
        const char* bb = p.bias ? p.bias + (b * p.bs[0] + h * p.bs[1]) * (BF32 ? 4 : 2) : p.q;
        const char* mb = (VMASK && p.mask) ? reinterpret_cast<const char*>(p.mask) + (b * p.ms[0] + h * p.ms[1]) : p.q;
        brs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(bb), 0, p.bias ? p.bias_bytes : 0u, 0x00020000);
        mrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(mb), 0, (VMASK && p.mask) ? p.mask_bytes : 0u, 0x00020000);
        abvo0 = (unsigned)((arow0 * p.bs[2] + akc * KPC) * (BF32 ? 4 : 2));
        amvo0 = (unsigned)(arow0 * p.ms[2] + akc * KPC);
    }
    auto add_gload = [&](int row0) {
        const int kcol0 = kblk * BN_;
#pragma unroll
        for (int i = 0; i < ACH; ++i) {
            stA[i] = __builtin_amdgcn_raw_buffer_load_b128(brs, abvo0, ((row0 + i * RSTEP) * (int)p.bs[2] + kcol0) * (BF32 ? 4 : 2), 0);
            if (VMASK && BF32) stM[i] = u32x2{__builtin_amdgcn_raw_buffer_load_b32(mrs, amvo0, (row0 + i * RSTEP) * (int)p.ms[2] + kcol0, 0), 0u};
            else if (VMASK) stM[i] = __builtin_amdgcn_raw_buffer_load_b64(mrs, amvo0, (row0 + i * RSTEP) * (int)p.ms[2] + kcol0, 0);
            else stM[i] = u32x2{0u, 0u};
        }
    };
    auto add_lstore = [&](int buf) {
#pragma unroll
        for (int i = 0; i < ACH; ++i) {
            u32x4 o;
            if constexpr (BF32) {
#pragma unroll
                for (int w = 0; w < 4; ++w) o[w] = (((stM[i][0] | nomask) >> (8 * w)) & 0xffu) ? stA[i][w] : 0xFF800000u;   // -inf where the key's mask byte is clear
                *LDS_PTR(u32x4, ldsAdd + (ADD_BUFS == 2 ? buf : 0) * ADDB + (arow0 + i * RSTEP) * (BN_ * 4) + akc * 16) = o;
                continue;
            }
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                const uint32_t mw = (stM[i][w >> 1] | nomask) >> (16 * (w & 1));   // mask bytes of keys 2w, 2w+1
                const uint32_t lo = (mw & 0xffu) ? (stA[i][w] & 0xffffu) : ninf16;
                const uint32_t hi16 = (mw & 0xff00u) ? (stA[i][w] >> 16) : ninf16;
                o[w] = lo | (hi16 << 16);
            }
            *LDS_PTR(u32x4, ldsAdd + (ADD_BUFS == 2 ? buf : 0) * ADDB + (akc >> 4) * (QT * 256) + tile_off<128>(arow0 + i * RSTEP, akc & 15)) = o;
        }
    };
    if (VEC && tq0 < ntq) {
        add_gload(tq0 * QT);
        add_lstore(0);
    }
    if (DIRECT) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
#pragma unroll
    for (int kb = 0; kb < KB; ++kb)
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            retire_loads(kf[kb][s]);
            retire_loads(vf[kb][s]);
            if (SEED_S) {
                uint16_t hk[8];
                __builtin_memcpy(hk, &kf[kb][s], 16);
                f32x8 f;
#pragma unroll
                for (int e = 0; e < 8; ++e) f[e] = E::to_f32(hk[e]) * p.c;
                kf[kb][s] = E::cvt8(f);
            }
        }

    constexpr bool KPD = mode_has_keypad(MODE);
    bool kp_keep[KB];
    bool kp_none = false;
    if (KPD) {
        bool any = false;
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) {
            const int key = kw0 + kb * 32 + l31;
            kp_keep[kb] = key < p.Sk && (p.mask == nullptr || p.mask[b * p.ms[0] + h * p.ms[1] + key] != 0);
            any = any || kp_keep[kb];
        }
        kp_none = !__any(any);
    }
    // the loop is unrolled by its two LDS buffers: the buffer offset is a compile-time constant and folds into the ds_read
    // immediates instead of costing VALU adds per LDS address (FASN_BWD_UNROLL2 = 0: dynamic buffer index, for A/B)
    using Buf0 = std::integral_constant<int, 0>;
    using Buf1 = std::integral_constant<int, FASN_BWD_UNROLL2 ? 1 : 0>;
    auto qtile_body = [&](const int tq, auto BUF_) {
        const int buf = FASN_BWD_UNROLL2 ? decltype(BUF_)::value : ((tq - tq0) & 1);
        const int r0 = tq * QT;
        if (tq + 1 < ntq) {
            if (DIRECT) {   // buffer buf^1 was released by the barrier that ended the previous tile
                tdQ.dma(qrw, ldsQ_w + (buf ^ 1) * TILEB, r0 + QT, p.qs[2]);
                tdD.dma(drw, ldsDO_w + (buf ^ 1) * TILEB, r0 + QT, bp.dos[2]);
            } else {
                tsQ.gload(stQ, qrs, r0 + QT, p.qs[2]);
                tsD.gload(stD, drs, r0 + QT, bp.dos[2]);
            }
            stats_gload(r0 + QT);
            if (VEC) add_gload(r0 + QT);
        }
        const char* tQ = ldsQ + buf * TILEB;
        const char* tD = ldsDO + buf * TILEB;
        const char* tA = ldsAdd + (ADD_BUFS == 2 ? buf : 0) * ADDB;
        const float* tL = ldsLse + buf * QT;
        const float* tX = ldsDlt + buf * QT;

        // wave-uniform classification of (this q tile) x (this wave's keys [kw0, kw0+KB*32))
        bool skip = false, need_mask = false;
        if (causal) {
            skip = (r0 + QT - 1 + coff) < kw0;                       // even the last row sees none of my keys
            need_mask = (r0 + coff) < (kw0 + KB * 32 - 1);           // the first row does not see all my keys
        }
        if (r0 + QT > p.Sq || kw0 + KB * 32 > p.Sk) need_mask = true;
        if (MODE == MODE_GENERAL_SLOW) need_mask = true;
        if (KPD && kp_none) skip = true;   // none of this wave's keys is visible to anybody

        if (!skip) {
#pragma unroll
            for (int qb = 0; qb < 2; ++qb) {
                // per-row statistics for the 16 rows this lane's registers cover (negated where they seed the accumulators)
                f32x16 lr, xr;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    f32x4 a = *LDS_PTR(const f32x4, tL + qb * 32 + 8 * g + 4 * hi);
                    f32x4 c = *LDS_PTR(const f32x4, tX + qb * 32 + 8 * g + 4 * hi);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        lr[4 * g + e] = a[e];
                        xr[4 * g + e] = c[e];
                    }
                }
                // S[q][key], dP[q][key] for 32 rows x KB*32 keys
                f32x16 sacc[KB], pacc[KB];
#pragma unroll
                for (int kb = 0; kb < KB; ++kb) {
                    if (VEC) {   // (all lanes take part in the transposed read; a lane whose key is padded overrides its values below)
                        const int cb = wave * KB + kb;   // this wave's 32-key column block inside the additive tile
                        if constexpr (BF32) {   // register r = row (r&3) + 8(r>>2) + 4hi of the 32-row block, column = this lane's key
#pragma unroll
                            for (int r = 0; r < 16; ++r) {
                                const float av = *LDS_PTR(const float, tA + (qb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi) * (BN_ * 4) + (cb * 32 + l31) * 4);
                                sacc[kb][r] = __builtin_fmaf(av, binv, lr[r]);
                            }
                        } else
#pragma unroll
                        for (int t2 = 0; t2 < 2; ++t2) {
                            vec8 av = lds_read_trfrag<E, 128>(tA + (cb >> 2) * (QT * 256), qb * 32 + 16 * t2, cb & 3, lane);
                            uint16_t ab[8];
                            __builtin_memcpy(ab, &av, 16);
#pragma unroll
                            for (int e = 0; e < 8; ++e) sacc[kb][8 * t2 + e] = __builtin_fmaf(E::to_f32(ab[e]), binv, lr[8 * t2 + e]);
                        }
                        if (KPD) {   // bias + key padding: a lane's key is hidden for every row
#pragma unroll
                            for (int r = 0; r < 16; ++r) sacc[kb][r] = kp_keep[kb] ? sacc[kb][r] : -INFINITY;
                        }
                    } else if (KPD && !kp_keep[kb]) {   // key-padding: a lane's key is hidden for every row
#pragma unroll
                        for (int r = 0; r < 16; ++r) sacc[kb][r] = -INFINITY;
                    } else {
#pragma unroll
                        for (int r = 0; r < 16; ++r) sacc[kb][r] = SEED_S ? lr[r] : 0.f;
                    }
#pragma unroll
                    for (int r = 0; r < 16; ++r) pacc[kb][r] = SEED_P ? xr[r] : 0.f;
                }
                const int lane_s = (D == 256 && MODE == MODE_GENERAL_SLOW && FASN_DKDV256_FRESH) ? fresh_lane_id() : lane;   // (likewise the row fragments' addresses)
#pragma unroll
                for (int s = 0; s < KS; ++s) {
                    vec8 qa = lds_read_rowfrag<E, D>(tQ, qb * 32 + (lane_s & 31), s, lane_s >> 5);
                    vec8 da = lds_read_rowfrag<E, D>(tD, qb * 32 + (lane_s & 31), s, lane_s >> 5);
#pragma unroll
                    for (int kb = 0; kb < KB; ++kb) {
                        sacc[kb] = E::mfma(qa, kf[kb][s], sacc[kb]);
                        pacc[kb] = E::mfma(da, vf[kb][s], pacc[kb]);
                    }
                }
                vec8 pfr[KB][2], dsfr[KB][2];
#pragma unroll
                for (int kb = 0; kb < KB; ++kb) {
                    const int key = kw0 + kb * 32 + l31;
                    auto elems = [&](auto MASKED) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        float y = sacc[kb][r] * p.c;
                        bool show = true;
                        if (decltype(MASKED)::value) {
                            const int row = r0 + qb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                            show = (key < p.Sk) && (row < p.Sq) && (!causal || key <= row + coff);
                            if (MODE == MODE_GENERAL_SLOW) {
                                if (p.bias != nullptr && show) {
                                    const int64_t bo = b * p.bs[0] + h * p.bs[1] + (int64_t)row * p.bs[2] + (int64_t)key * p.bs[3];
                                    float bv;
                                    if (p.bias_f32) bv = reinterpret_cast<const float*>(p.bias)[bo];
                                    else bv = E::to_f32(reinterpret_cast<const uint16_t*>(p.bias)[bo]);
                                    y = __builtin_fmaf(bv, kLog2e, y);
                                }
                                if (p.mask != nullptr && show) {
                                    const int64_t mo = b * p.ms[0] + h * p.ms[1] + (int64_t)row * p.ms[2] + (int64_t)key * p.ms[3];
                                    show = p.mask[mo] != 0;
                                }
                            }
                        }
                        float pv = (MODE == MODE_GENERAL_SLOW) ? fast_exp2(y - lr[r]) : fast_exp2(sacc[kb][r]);
                        if (decltype(MASKED)::value) pv = show ? pv : 0.f;
                        float dp = pacc[kb][r];
                        float pd = pv;
                        if (DROP) {   // lane = key here: the state of the register's row comes from the lane of the key quad that computed it
                            // (the state of row 8 g + 4 hi + (lane & 3) of the block and this lane's key group: one per register group g - CSE - fetched from
                            // the lane of the key quad that computed it, fasn_common.h: quad_bcast)
                            const uint32_t own = drop_mix(drop_row_base(dsd.lo, (uint32_t)bh, (uint32_t)(r0 + qb * 32 + 8 * (r >> 2) + 4 * hi + (lane & 3))), dsd.hi, (uint32_t)(key >> 4));
                            const bool keep = (int32_t)drop_word(quad_bcast(own, r & 3), dlane) >= dthr.hi32;
                            const float ks = keep ? p.drop_scale : 0.f;   // (one select: the factor, 1 / (1 - p) or 0)
                            dp *= ks;
                            pd = pv * ks;
                        }
                        sacc[kb][r] = pd;                     // dropped weights feed dV
                        pacc[kb][r] = SEED_P ? pv * dp : pv * (dp - xr[r]);      // dS uses the undropped P
                    }
                    };
                    // (dropout instantiations: ONE copy of the element pass - the masked one)
                    if (DROP || need_mask) elems(std::true_type{});
                    else elems(std::false_type{});
#pragma unroll
                    for (int t2 = 0; t2 < 2; ++t2) {
                        f32x8 x, y;
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            x[e] = sacc[kb][8 * t2 + e];
                            y[e] = pacc[kb][8 * t2 + e];
                        }
                        pfr[kb][t2] = E::cvt8(x);
                        dsfr[kb][t2] = E::cvt8(y);
                    }
                }
                // dV^T[d][key] += dO^T[d][q] P[q][key];  dK^T[d][key] += Q^T[d][q] dS[q][key]
                // (D = 256 vector modes: the addresses of the transposed reads from a fresh lane id - kept across the tile loop, 43 - 59 of them went to scratch)
                const int lane_t = (D == 256 && (VEC || MODE == MODE_GENERAL_SLOW) && FASN_DKDV256_FRESH) ? fresh_lane_id() : lane;
#pragma unroll
                for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
                    for (int d = 0; d < DB; ++d) {
                        vec8 dot = lds_read_trfrag<E, D>(tD, qb * 32 + 16 * t2, d0 + d, lane_t);
                        vec8 qt = lds_read_trfrag<E, D>(tQ, qb * 32 + 16 * t2, d0 + d, lane_t);
#pragma unroll
                        for (int kb = 0; kb < KB; ++kb) {
                            dvacc[kb][d] = E::mfma(dot, pfr[kb][t2], dvacc[kb][d]);
                            dkacc[kb][d] = E::mfma(qt, dsfr[kb][t2], dkacc[kb][d]);
                        }
                    }
            }
        }
        if (tq + 1 < ntq) {
            if (!DIRECT) {
                tsQ.lstore(stQ, ldsQ + (buf ^ 1) * TILEB);
                tsD.lstore(stD, ldsDO + (buf ^ 1) * TILEB);
            }
            stats_lstore(buf ^ 1);
            if (VEC) {
                if (ADD_BUFS == 1) __syncthreads();   // every wave is done with the tile this one replaces (tq, ntq: the same for the whole workgroup)
                add_lstore(buf ^ 1);
            }
        }
        if (DIRECT) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the next Q / dO tiles have landed
        __syncthreads();
    };
    if constexpr (FASN_BWD_UNROLL2 != 0) {
        for (int tq = tq0; tq < ntq; tq += 2) {
            qtile_body(tq, Buf0{});
            if (tq + 1 < ntq) qtile_body(tq + 1, Buf1{});
        }
    } else {
        for (int tq = tq0; tq < ntq; ++tq) qtile_body(tq, Buf0{});
    }

    }   // query heads of the group

    char* dkbase = bp.dk + (b * bp.dks[0] + hk * bp.dks[1]) * 2;   // dK / dV are [B, H / kvg, Sk, D]
    char* dvbase = bp.dv + (b * bp.dvs[0] + hk * bp.dvs[1]) * 2;
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) {
        const int key = kw0 + kb * 32 + l31;
        if (key < p.Sk) {
            char* rk = dkbase + (int64_t)key * bp.dks[2] * 2;
            char* rv = dvbase + (int64_t)key * bp.dvs[2] * 2;
#pragma unroll
            for (int d = 0; d < DB; ++d) {   // (8-byte stores: several instantiations at their register limit, fasn_common.h)
                store_block_narrow<E>(rk + (d0 + d) * 64, dkacc[kb][d], bp.scale, hi);
                store_block_narrow<E>(rv + (d0 + d) * 64, dvacc[kb][d], 1.0f, hi);
            }
        }
    }
    }   // pass
}

}  // namespace fasn

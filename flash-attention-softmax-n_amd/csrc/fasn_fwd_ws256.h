// fasn_fwd_ws256.h — forward at head dim 256 with the two GEMMs of a score block split over TWO cooperating waves of one SIMD
// (the forward counterpart of fasn_bwd_dq_ws.h; reference math: core/functional.py:15-29,32-93, FA-2 recurrence with the "+n" sink).
//
// Round 3 served D = 256 with one wave per SIMD and "feature halves": Q^T fragments (64 registers) plus a full O^T accumulator (128)
// do not fit next to the score tile, so two workgroups shared a row block, each computed QK^T in full and owned half of the output
// features - 3 GEMM-equivalents for 2, one wave per SIMD, 0.15 of the MFMA peak. Here a workgroup has 8 waves for 128 query rows; waves
// w and w + 4 share a SIMD and the same 32 rows (a lane owns one query row, as everywhere in the forward):
//
//   wave A (w < 4):  S^T = K Q'^T  ->  online softmax_n (running max m, sum l)  ->  P^T (16 bit) and the rescale factor to LDS
//   wave B (w >= 4): reads P^T / alpha of the previous key block  ->  O^T = alpha O^T + V^T P^T
//
// A holds Q' (64 registers) and the statistics, B the O^T accumulator (128): both fit 256 registers, two waves per SIMD, 16 MFMAs
// each per 32-key block, no score computed twice. B runs one key block behind A: the P^T of block u is published by the barrier
// that ends iteration u. K / V arrive by LDS-DMA in units of 32 keys (16 KiB images [32][256], swizzled like every tile), K and V in rings
// of four, both requested two blocks ahead of their first reader (vmcnt counts in order: a deeper K ring behind a shallow V ring would
// be drained by the wait for V anyway); one barrier per block. Plain, causal and key-padding launches (MODE_KEYPAD: wave A takes
// the 32 visibility bits of a key block from a per-workgroup word table and the blocks behind the last visible key are not walked).
// MODE_GENERAL (round 6: a dense boolean mask and / or a 16-bit additive bias whose rows move as vectors; either may be absent): wave A keeps a
// private ring of two images of its [32 rows][32 keys] - 2 KiB of bias, 1 KiB of mask bytes, three LDS-DMA requests, asked for two blocks
// ahead right after the block's image has been read into registers - and starts its score accumulator at bias*log2e (-inf where the mask
// byte is clear) instead of zero; nothing else changes, wave B does not know. The 24 KiB of images are paid for with the fourth V slot
// (V is then requested one block ahead of the block wave B is working on). Dropout stays on the feature-half kernels.
#pragma once
#include "fasn_fwd_kernel.h"

namespace fasn {

constexpr int W256_NK = 4, W256_NV = 4;           // K / V ring slots
constexpr int W256_UNIT = 32 * 256 * 2;           // one 32-key image
constexpr int W256_IMG = 3072;                    // MODE_GENERAL: one image slot of an A wave (2 KiB bias + 1 KiB mask)
constexpr int ws256_smem_bytes(int mode = MODE_KEYPAD) {
    return mode == MODE_GENERAL ? (W256_NK + 3) * W256_UNIT + 2 * 4 * 2048 + 2 * 4 * 256 + 4 * 2 * W256_IMG
                                : (W256_NK + W256_NV) * W256_UNIT + 2 * 4 * 2048 + 2 * 4 * 256 + kFwdKpMaxTiles * 8;
}

// DROP = 1 (MODE_GENERAL only - it serves every dropout call at this head dim, operands or not): wave A clears the dropped weights in the
// PACKED pairs it hands to wave B (stream definition 2, fasn_common.h: DropBlock, the plain register layout); its row sums keep the undropped
// weights and 1 / (1 - p) goes into the final 1 / l, as in fasn_fwd_kernel.h.
template <typename Tag, int MODE, int DROP = 0>
__global__ void __launch_bounds__(512, 2) fasn_fwd_ws256_kernel(const FwdParams p) {
    static_assert(!DROP || MODE == MODE_GENERAL, "two-wave D = 256 forward: dropout through the general instantiation");
    static_assert(MODE == MODE_PLAIN || MODE == MODE_CAUSAL || MODE == MODE_KEYPAD || MODE == MODE_GENERAL, "two-wave D = 256 forward: plain, causal, key padding, vector mask / bias");
    using E = ET<Tag>;
    using vec8 = typename E::vec8;
    constexpr int D = 256, KS = 16, DB = 8, BM = 128, KU = 32;
    constexpr bool KP = MODE == MODE_KEYPAD;   // a boolean mask over (batch, head, key), with or without the causal flag
    constexpr bool GEN = MODE == MODE_GENERAL; // dense mask and / or 16-bit bias as per-wave LDS images
    constexpr int NV = GEN ? 3 : W256_NV;      // V ring slots
    constexpr int NIMG = GEN ? 3 : 0;          // image requests of an A wave per block
    const bool causal = MODE == MODE_CAUSAL || ((KP || GEN) && p.causal != 0);

    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const ldsK = smem;                                        // [W256_NK][UNIT]
    char* const ldsV = smem + W256_NK * W256_UNIT;                  // [W256_NV][UNIT]
    char* const ldsP = smem + (W256_NK + NV) * W256_UNIT;           // [2][4 row blocks][2 KiB]
    float* const ldsA = reinterpret_cast<float*>(ldsP + 2 * 4 * 2048);   // [2][4][64] rescale factor of the lane's row
    uint64_t* const ldsKP = reinterpret_cast<uint64_t*>(ldsP + 2 * 4 * 2048 + 2 * 4 * 256);   // [kFwdKpMaxTiles] visibility words of 64 keys (key-padding mode)
    char* const ldsI = ldsP + 2 * 4 * 2048 + 2 * 4 * 256;           // [4 A waves][2][W256_IMG] bias + mask images (MODE_GENERAL; no visibility words there)

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31;
    const int hi = lane >> 5;
    const int role = wave >> 2;   // 0 = A, 1 = B
    const int rbw = wave & 3;

    int bh, qi;
    block_to_work_grouped((int)blockIdx.x, p.B * p.H, p.nqblk, (FASN_CAUSAL_GROUPS && causal) ? causal_head_group(p.B * p.H, p.Sk, D) : 1, bh, qi);   // (causal: heads in groups, fasn_common.h)
    const int qblk = causal ? (p.nqblk - 1 - qi) : qi;   // heaviest blocks first
    const int b = bh / p.H, h = bh % p.H;
    const int q0 = qblk * BM;
    const int qw0 = q0 + rbw * 32;
    const int row = qw0 + l31;
    const bool row_ok = row < p.Sq;
    const int coff = p.Sk - p.Sq;
    const int vis = causal ? row + coff : 0x7fffffff;

    const char* kbase = p.k + (b * p.ks[0] + (h / p.kvg) * p.ks[1]) * 2;
    const char* vbase = p.v + (b * p.vs[0] + (h / p.kvg) * p.vs[1]) * 2;

    int nu = (p.Sk + KU - 1) / KU;   // key blocks this workgroup walks
    const int nu_all = nu;           // the first unit behind the last key: a request for it is out of the descriptors' range (zero fill, no traffic)
    if (causal) {
        const int kmax = min(q0 + BM, p.Sq) - 1 + coff;
        nu = min(nu, kmax < 0 ? 0 : kmax / KU + 1);
    }

    // ---- K / V units straight to LDS: thread `tid` owns 16-byte slots tid and tid + 512 of an image
    unsigned voffK[2], voffV[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int ci = tid + i * 512;
        const int r = ci >> 5, ch = (ci & 31) ^ swz_f<D>(r);
        voffK[i] = (unsigned)(r * (int)p.ks[2] * 2 + ch * 16);
        voffV[i] = (unsigned)(r * (int)p.vs[2] * 2 + ch * 16);
    }
    const u32x4 krw = make_rsrc_words(kbase, p.kbytes), vrw = make_rsrc_words(vbase, p.vbytes);
    const uint32_t ldsK_w = lds_addr(ldsK) + wave * 1024, ldsV_w = lds_addr(ldsV) + wave * 1024;
    auto k_dma = [&](int u, int slot) {   // (units past the end of K read back as zeros; 2 requests)
#pragma unroll
        for (int i = 0; i < 2; ++i) lds_dma16(krw, __builtin_amdgcn_readfirstlane(ldsK_w + slot * W256_UNIT + i * 8192), voffK[i], (uint32_t)u * (uint32_t)(KU * (int)p.ks[2] * 2));
    };
    auto v_dma = [&](int u, int slot) {
#pragma unroll
        for (int i = 0; i < 2; ++i) lds_dma16(vrw, __builtin_amdgcn_readfirstlane(ldsV_w + slot * W256_UNIT + i * 8192), voffV[i], (uint32_t)u * (uint32_t)(KU * (int)p.vs[2] * 2));
    };

    // ---- MODE_GENERAL: the images of this A wave. Bias: [32 rows][64 B] as 128 pieces of 16 bytes (8 keys), piece chunk ^ swz_f<32>(row) of a row
    // in place `chunk` (the forward's dense-mask image layout); mask: [32 rows][32 B] row major, 64 pieces. A lane owns a ROW: its accumulator
    // registers 4g .. 4g+3 are keys 8g + 4hi + 0..3 of the block - 8 bytes of the bias row, 4 of the mask row.
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
    char* const img = ldsI + rbw * (2 * W256_IMG);
    const uint32_t img_a = lds_addr(img);
    auto img_dma = [&](int u, int slot) {   // 3 requests; blocks past the last key are out of the descriptors' range: zeros, no traffic
#pragma unroll
        for (int i = 0; i < 2; ++i) lds_dma16(brw, __builtin_amdgcn_readfirstlane(img_a + slot * W256_IMG + i * 1024), bvo[i], (uint32_t)u * (uint32_t)(KU * 2));
        lds_dma16(mrw, __builtin_amdgcn_readfirstlane(img_a + slot * W256_IMG + 2048), mvo, (uint32_t)u * (uint32_t)KU);
    };
    const uint32_t nomask = (GEN && p.mask == nullptr) ? 0x01010101u : 0u;   // no mask operand: every byte reads as set
    const char* const img_rd_b = img + l31 * 64 + hi * 8;   // + ((g ^ swz) << 4)
    const char* const img_rd_m = img + 2048 + l31 * 32 + hi * 4;   // + 8 g
    const int img_swz = swz_f<32>(l31);

    // ---- prologue: the first K / V units (iteration u requests K unit u + 3 and V unit u + 2; before the loop: K 0..2, V 0..1.
    // MODE_GENERAL, three V slots: iteration u requests V unit u + 1; before the loop: V 0, and the images of blocks 0 and 1)
#pragma unroll
    for (int u = 0; u < W256_NK - 1; ++u) k_dma(u, u);
    v_dma(0, 0);
    if (!GEN) v_dma(1, 1);
    if (GEN && role == 0) {
        img_dma(0, 0);
        img_dma(1, 1);
    }
    // softmax_n state of the lane's row: the sink column (logit 0, weight n) is the start value; each half-wave sums its own 16 keys
    // per block and the halves are merged at the end (the maximum is shared every block, so both halves scale alike)
    float m_run = p.n > 0.f ? 0.f : -INFINITY;
    float l_run = (p.n > 0.f && hi == 0) ? p.n : 0.f;
    const int wave_first_vis = qw0 + coff, wave_last_vis = qw0 + 31 + coff;
    // wave-uniform classification of (this wave's 32 rows) x (key block u): identical for the A and the B wave of a row block
    auto classify = [&](int u, bool& skip, bool& need_mask, uint32_t& kpb) {
        const int k0 = u * KU;
        skip = qw0 >= p.Sq;
        need_mask = k0 + KU > p.Sk;
        kpb = ~0u;
        if (causal) {
            skip = skip || k0 > wave_last_vis;
            need_mask = need_mask || (k0 + KU - 1) > wave_first_vis;
        }
        if (KP) {   // the 32 visibility bits of this key block
            kpb = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(ldsKP[u >> 1] >> (32 * (u & 1))));
            skip = skip || kpb == 0u;
            need_mask = need_mask || kpb != ~0u;
        }
    };
    const DropSeed dsd = DROP ? drop_seed(p.seed_lo, p.seed_hi, p.rng) : DropSeed{0u, 0u};
    const DropThr dthr = drop_thr(DROP ? p.drop_thr : 1u);
    const uint32_t drop_rh = drop_rh_of(hi);
    const uint32_t drop_rb = DROP ? drop_row_base(dsd.lo, (uint32_t)bh, (uint32_t)row) : 0u;
    char* const pslot = ldsP + rbw * 2048 + lane * 16;      // + parity * 8192 (+ 1024 for the second half of the block)
    float* const aslot = ldsA + rbw * 64 + lane;            // + parity * 256

    // ---- wave A: key block u
    u32x2 braw[4] = {};   // MODE_GENERAL: the lane's 16 bias values / 16 mask bytes of the block wave A is working on
    uint32_t mraw[4] = {};
    auto img_read = [&](int slot) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            braw[g] = *LDS_PTR(const u32x2, img_rd_b + slot * W256_IMG + ((g ^ img_swz) << 4));
            mraw[g] = *LDS_PTR(const uint32_t, img_rd_m + slot * W256_IMG + 8 * g);
        }
    };
    auto block_a = [&](const int u, const int kslot, const vec8 (&qf)[KS]) {
        bool skip, need_mask;
        uint32_t kpb;
        classify(u, skip, need_mask, kpb);
        char* ps = pslot + (u & 1) * 8192;
        if (skip) {   // nothing visible to these rows: B is told to leave its accumulator alone and gets zero weights
            *LDS_PTR(u32x4, ps) = u32x4{0u, 0u, 0u, 0u};
            *LDS_PTR(u32x4, ps + 1024) = u32x4{0u, 0u, 0u, 0u};
            aslot[(u & 1) * 256] = 1.0f;
            return;
        }
        const char* tK = ldsK + kslot * W256_UNIT;
        f32x16 sacc;
#pragma unroll
        for (int r = 0; r < 16; ++r) sacc[r] = 0.f;
        if constexpr (GEN) {   // start values from the block's image (the caller has read it: braw / mraw)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const uint32_t w = braw[r >> 2][(r & 3) >> 1];
                const float bv = E::to_f32((uint16_t)((r & 1) ? (w >> 16) : (w & 0xffffu))) * kLog2e;
                sacc[r] = (((mraw[r >> 2] | nomask) >> (8 * (r & 3))) & 0xffu) ? bv : -INFINITY;
            }
        }
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const vec8 kf = lds_read_rowfrag<E, D>(tK, l31, s, hi);
            sacc = E::mfma(kf, qf[s], sacc);
        }
        if (need_mask) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int kin = (r & 3) + 8 * (r >> 2) + 4 * hi, key = u * KU + kin;
                sacc[r] = (key < p.Sk && key <= vis && (!KP || ((kpb >> kin) & 1u))) ? sacc[r] : -INFINITY;
            }
        }
        float tmax = sacc[0];
#pragma unroll
        for (int r = 1; r < 16; ++r) tmax = fmaxf(tmax, sacc[r]);
        tmax = max_across_halves(tmax);
        const float m_new = fmaxf(m_run, tmax);
        const float m_use = m_new == -INFINITY ? 0.f : m_new;   // (a row that has seen no key yet: every weight exp2(-inf) = 0)
        const float alpha = fast_exp2(m_run - m_use);           // m_run = -inf: 0, and l / O are 0 there anyway
        m_run = m_new;
        float rs = 0.f;
        vec8 pfr[2];
#pragma unroll
        for (int t2 = 0; t2 < 2; ++t2) {
            f32x8 x;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                x[e] = fast_exp2(sacc[8 * t2 + e] - m_use);
                rs += x[e];
            }
            pfr[t2] = E::cvt8(x);
        }
        l_run = l_run * alpha + rs;
        u32x4 w0, w1;
        __builtin_memcpy(&w0, &pfr[0], 16);
        __builtin_memcpy(&w1, &pfr[1], 16);
        if constexpr (DROP != 0) {
            const DropBlock<false> db(drop_rb, dsd.hi, (uint32_t)((u * KU) >> 4), hi, drop_rh);
#pragma unroll
            for (int e = 0; e < 4; ++e) {   // dword e of half t2 = accumulator registers 8 t2 + 2 e, + 1
                w0[e] &= db.keep_mask_pk(e, dthr);
                w1[e] &= db.keep_mask_pk(4 + e, dthr);
            }
        }
        *LDS_PTR(u32x4, ps) = w0;
        *LDS_PTR(u32x4, ps + 1024) = w1;
        aslot[(u & 1) * 256] = alpha;
    };
    // ---- wave B: key block u (the one wave A finished in the previous iteration)
    auto block_b = [&](const int u, const int vslot, f32x16 (&oacc)[DB]) {
        const char* ps = pslot + (u & 1) * 8192;
        const u32x4 w0 = *LDS_PTR(const u32x4, ps), w1 = *LDS_PTR(const u32x4, ps + 1024);
        const float alpha = aslot[(u & 1) * 256];
        bool skip, need_mask;
        uint32_t kpb;
        classify(u, skip, need_mask, kpb);
        if (skip) return;
        if (__any(alpha != 1.0f)) {   // some row's running maximum moved (every row in the first blocks, rarely later)
#pragma unroll
            for (int d = 0; d < DB; ++d)
#pragma unroll
                for (int r = 0; r < 16; ++r) oacc[d][r] *= alpha;
        }
        vec8 pfr[2];
        __builtin_memcpy(&pfr[0], &w0, 16);
        __builtin_memcpy(&pfr[1], &w1, 16);
        const char* tV = ldsV + vslot * W256_UNIT;
#pragma unroll
        for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
            for (int d = 0; d < DB; ++d) {
                const vec8 vtf = lds_read_trfrag<E, D>(tV, 16 * t2, d, lane);
                oacc[d] = E::mfma(vtf, pfr[t2], oacc[d]);
            }
    };

    // iteration u: A works on K unit u, B on V unit u - 1; K unit u + 3 and V unit u + 2 are requested (their slots were released by
    // the barrier that ended iteration u - 1). K unit u + 1 and V unit u - needed in iteration u + 1 - were requested in iteration
    // u - 2, so the wait that ends an iteration leaves the requests of this iteration and the previous one in flight: 2 x (2 + 2) per wave.
    // Each role runs its own copy of the loop (same trip count, same barriers): Q' lives in A's branch only, O^T in B's.
    auto requests = [&](int u) {
        // (past the end of the WALK - which a causal bound or a key-padding trim may have shortened - ask for the unit behind the last KEY:
        // out of range, zero fill, no traffic; the request counts stay uniform)
        if constexpr (GEN) {
            // V first, K last: vmcnt counts in order, and what the barrier at the end of this iteration has to publish - V unit u, the images of
            // block u + 1 (both asked for in iteration u - 1) - then sits in front of K unit u + 2, which may stay in flight with this iteration's
            v_dma(u + 1 < nu ? u + 1 : nu_all, (u + 1) % 3);
            if (role == 0) {
                if (u < nu) img_read(u & 1);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the image is in registers before its slot is asked for again
                img_dma(u + 2 < nu ? u + 2 : nu_all, u & 1);
            }
            k_dma(u + 3 < nu ? u + 3 : nu_all, (u + 3) & 3);
        } else {
            k_dma(u + 3 < nu ? u + 3 : nu_all, (u + 3) & 3);
            v_dma(u + 2 < nu ? u + 2 : nu_all, (u + 2) & 3);
        }
    };
    auto close = [&](auto ROLE_) {
        if constexpr (GEN) {
            if constexpr (decltype(ROLE_)::value == 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(6 + NIMG) : "memory");
            else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        }
        __syncthreads();
    };
    using RoleA = std::integral_constant<int, 0>;
    using RoleB = std::integral_constant<int, 1>;
    if (KP) kp_build_words(ldsKP, p.mask ? p.mask + (b * p.ms[0] + h * p.ms[1]) : nullptr, p.Sk, (p.Sk + 63) / 64, tid, 512);   // published by the first barrier below
    auto trim = [&]() {   // key-padding: blocks behind the last visible key are not walked (every wave finds the same one)
        if (!KP) return;
        int last = -1;
        for (int t = lane; t < (p.Sk + 63) / 64; t += 64) {
            const uint64_t w = ldsKP[t];
            if (w != 0ull) last = 2 * t + ((w >> 32) != 0ull ? 1 : 0);
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) last = max(last, __shfl_xor(last, o));
        nu = min(nu, __builtin_amdgcn_readfirstlane(last + 1));
    };
    if (role == 0) {
        vec8 qf[KS];
        {
            const char* rq = p.q + (b * p.qs[0] + h * p.qs[1] + (int64_t)row * p.qs[2]) * 2 + hi * 16;
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                u32x4 a = {0u, 0u, 0u, 0u};
                if (row_ok) a = gload16(rq + s * 32);
                __builtin_memcpy(&qf[s], &a, 16);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        trim();
#pragma unroll
        for (int s = 0; s < KS; ++s) {   // Q' = Q * scale*log2e, rounded to the operand type (as core/flash_attn.py:81-83 does with its pre-scaled q)
            retire_loads(qf[s]);
            uint16_t hq[8];
            __builtin_memcpy(hq, &qf[s], 16);
            f32x8 f;
#pragma unroll
            for (int e = 0; e < 8; ++e) f[e] = E::to_f32(hq[e]) * p.c;
            qf[s] = E::cvt8(f);
        }
        for (int u = 0; u <= nu; ++u) {
            requests(u);
            if (u < nu) block_a(u, u & 3, qf);
            close(RoleA{});
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const float l_tot = sum_across_halves(l_run);
        aslot[0] = l_tot > 0.f ? (DROP ? p.drop_scale : 1.0f) / l_tot : 0.f;
        if (row_ok && p.lse != nullptr && hi == 0) {
            const float m_use = (m_run == -INFINITY) ? 0.f : m_run;
            p.lse[(int64_t)bh * p.Sq + row] = l_tot > 0.f ? (m_use + __builtin_log2f(l_tot)) * kLn2 : -INFINITY;
        }
        __syncthreads();
    } else {
        f32x16 oacc[DB];
#pragma unroll
        for (int d = 0; d < DB; ++d)
#pragma unroll
            for (int r = 0; r < 16; ++r) oacc[d][r] = 0.f;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        trim();
        for (int u = 0; u <= nu; ++u) {
            requests(u);
            if (u > 0) block_b(u - 1, GEN ? (u - 1) % 3 : (u - 1) & 3, oacc);
            close(RoleB{});
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();   // 1 / l of every row is published
        if (row_ok) {
            const float inv = aslot[0];
            char* rp = p.o + (b * p.os[0] + h * p.os[1] + (int64_t)row * p.os[2]) * 2;
#pragma unroll
            for (int d = 0; d < DB; ++d) store_block_wide<E>(rp + d * 64, oacc[d], inv, hi);   // 16-byte stores (round 5, fasn_common.h)
        }
    }
}

}  // namespace fasn

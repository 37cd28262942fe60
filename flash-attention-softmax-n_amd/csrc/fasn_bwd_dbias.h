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
// K / V rows are staged key-permuted (as in the vector mask / bias kernels) so that a lane's 16 accumulator registers of a
// 32-key block are 16 CONSECUTIVE keys: bias, mask and dbias rows are then 32 / 16 / 32 contiguous bytes per lane.
#pragma once
#include "fasn_bwd_kernel.h"

namespace fasn {

struct DbiasParams {
    BwdParams b;           // q,k,v,lse,dout,delta,mask,bias (+ strides, sizes, scale, causal); b.dbias = output base
    int Bb, Hb;            // extent of the bias over batch / heads: 1 = broadcast (reduce over it), else B / H
    int out_f32;           // dbias elements are fp32 (else the 16-bit type of q)
    int nqb, nkb;          // 128-row / 128-key blocks
};

template <typename Tag, int D>
__global__ void __launch_bounds__(256, 1) fasn_bwd_dbias_kernel(const DbiasParams dp) {
    using E = ET<Tag>;
    using vec8 = typename E::vec8;
    const BwdParams& bp = dp.b;
    const FwdParams& p = bp.f;
    constexpr int KS = D / 16;
    constexpr int TILEB = KT * D * 2;
    constexpr int NLD = (KT * (D / 8)) / 256;

    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const ldsK = smem;            // [TILEB]
    char* const ldsV = smem + TILEB;    // [TILEB]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31;
    const int hi = lane >> 5;

    int blk = blockIdx.x;
    const int kblk = blk % dp.nkb; blk /= dp.nkb;
    const int qblk = blk % dp.nqb; blk /= dp.nqb;
    const int hb = blk % dp.Hb;
    const int bb = blk / dp.Hb;
    const int row = qblk * 128 + wave * 32 + l31;
    const bool row_ok = row < p.Sq;
    const int key0 = kblk * 128;
    const int coff = p.Sk - p.Sq;
    const bool causal = p.causal != 0;

    f32x16 dsum[2][2];   // [64-key tile][32-key block]: keys key0 + 64 t + 32 kb + 16 hi + r
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) dsum[t][kb][r] = 0.f;

    // the whole tile lies above the diagonal for every row of the workgroup: nothing to add (zeros are stored)
    const bool tile_hidden = causal && key0 > (min(qblk * 128 + 127, p.Sq - 1) + coff);

    const int b_lo = dp.Bb == 1 ? 0 : bb, b_hi = dp.Bb == 1 ? p.B : bb + 1;
    const int h_lo = dp.Hb == 1 ? 0 : hb, h_hi = dp.Hb == 1 ? p.H : hb + 1;
    TileStage<D, NLD> tsK, tsV;
    tsK.init(tid, p.ks[2], true);
    tsV.init(tid, p.vs[2], true);

    if (!tile_hidden)
    for (int b = b_lo; b < b_hi; ++b)
    for (int h = h_lo; h < h_hi; ++h) {
        const int bh = b * p.H + h;
        const char* qbase = p.q + (b * p.qs[0] + h * p.qs[1]) * 2;
        const char* kbase = p.k + (b * p.ks[0] + (h / p.kvg) * p.ks[1]) * 2;
        const char* vbase = p.v + (b * p.vs[0] + (h / p.kvg) * p.vs[1]) * 2;
        const char* dobase = bp.dout + (b * bp.dos[0] + h * bp.dos[1]) * 2;
        const __amdgpu_buffer_rsrc_t krs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(kbase), 0, p.kbytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t vrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(vbase), 0, p.vbytes, 0x00020000);
        // Q (pre-scaled by c = scale*log2e, rounded to the operand type as in every vector kernel) and dO fragments of this lane's row
        vec8 qf[KS], dof[KS];
        {
            const char* rq = qbase + (int64_t)row * p.qs[2] * 2 + hi * 16;
            const char* rd = dobase + (int64_t)row * bp.dos[2] * 2 + hi * 16;
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                u32x4 a = {0u, 0u, 0u, 0u}, d = {0u, 0u, 0u, 0u};
                if (row_ok) {
                    a = gload16(rq + s * 32);
                    d = gload16(rd + s * 32);
                }
                uint16_t hq[8];
                __builtin_memcpy(hq, &a, 16);
                f32x8 f;
#pragma unroll
                for (int e = 0; e < 8; ++e) f[e] = E::to_f32(hq[e]) * p.c;
                qf[s] = E::cvt8(f);
                __builtin_memcpy(&dof[s], &d, 16);
            }
        }
        const float l = row_ok ? p.lse[(int64_t)bh * p.Sq + row] : INFINITY;
        const float nlse2 = (l == -INFINITY || l == INFINITY) ? -INFINITY : -l * kLog2e;   // a row without weights: P = 0
        const float ndlt = row_ok ? -bp.delta[(int64_t)bh * p.Sq + row] : 0.f;
        const char* brow = p.bias + (b * p.bs[0] + h * p.bs[1] + (int64_t)row * p.bs[2]) * (p.bias_f32 ? 4 : 2);
        const uint8_t* mrow = p.mask ? p.mask + (b * p.ms[0] + h * p.ms[1] + (int64_t)row * p.ms[2]) : nullptr;
        const int vis = causal ? row + coff : 0x7fffffff;

#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int k0 = key0 + t * KT;
            if (k0 >= p.Sk || (causal && k0 > (min(qblk * 128 + 127, p.Sq - 1) + coff))) continue;   // (workgroup-uniform)
            u32x4 stK[NLD], stV[NLD];
            tsK.gload(stK, krs, k0, p.ks[2]);
            tsV.gload(stV, vrs, k0, p.vs[2]);
            __syncthreads();   // the previous tile's fragments have been read by every wave
            tsK.lstore(stK, ldsK);
            tsV.lstore(stV, ldsV);
            __syncthreads();
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
                const int kbase_key = k0 + kb * 32 + 16 * hi;   // register r = key kbase_key + r (key-permuted rows)
                f32x16 sacc, pacc;
                // start values: S' = bias*log2e - LSE*log2e (+ q'.k), dP' = -delta (+ dO.v)
                const bool full = row_ok && kbase_key + 16 <= p.Sk;
                if (full && p.bias_vec && !p.bias_f32) {   // 16 consecutive keys = 32 bytes of this lane's bias row (rows 8-byte aligned)
                    u32x2 w[4];
#pragma unroll
                    for (int g = 0; g < 4; ++g) w[g] = *reinterpret_cast<const u32x2*>(brow + (kbase_key + 4 * g) * 2);
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const uint32_t x = w[r >> 2][(r & 3) >> 1];
                        sacc[r] = __builtin_fmaf(E::to_f32((uint16_t)((r & 1) ? (x >> 16) : (x & 0xffffu))), kLog2e, nlse2);
                    }
                } else {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        float bv = 0.f;
                        const int key = kbase_key + r;
                        if (row_ok && key < p.Sk) {
                            if (p.bias_f32) bv = reinterpret_cast<const float*>(brow)[(int64_t)key * p.bs[3]];
                            else bv = E::to_f32(reinterpret_cast<const uint16_t*>(brow)[(int64_t)key * p.bs[3]]);
                        }
                        sacc[r] = __builtin_fmaf(bv, kLog2e, nlse2);
                    }
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) pacc[r] = ndlt;
                uint32_t mw[4] = {0x01010101u, 0x01010101u, 0x01010101u, 0x01010101u};   // mask bytes of the 16 keys (no mask: all visible)
                if (mrow != nullptr) {
                    if (full && p.mask_vec) {
#pragma unroll
                        for (int g = 0; g < 4; ++g) mw[g] = *reinterpret_cast<const uint32_t*>(mrow + kbase_key + 4 * g);
                    } else {
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            mw[g] = 0u;
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                const int key = kbase_key + 4 * g + e;
                                if (row_ok && key < p.Sk && mrow[(int64_t)key * p.ms[3]] != 0) mw[g] |= 1u << (8 * e);
                            }
                        }
                    }
                }
#pragma unroll
                for (int s = 0; s < KS; ++s) {
                    const vec8 kf = lds_read_rowfrag<E, D>(ldsK, kb * 32 + l31, s, hi);
                    const vec8 vf = lds_read_rowfrag<E, D>(ldsV, kb * 32 + l31, s, hi);
                    sacc = E::mfma(kf, qf[s], sacc);
                    pacc = E::mfma(vf, dof[s], pacc);
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = kbase_key + r;
                    const bool show = row_ok && key < p.Sk && key <= vis && ((mw[r >> 2] >> (8 * (r & 3))) & 0xffu) != 0;
                    const float pv = show ? fast_exp2(sacc[r]) : 0.f;
                    dsum[t][kb][r] = __builtin_fmaf(pv, pacc[r], dsum[t][kb][r]);
                }
            }
        }
    }

    // ---- store: 16 consecutive keys per lane and 32-key block
    if (!row_ok) return;
    char* orow = bp.dbias + (bb * bp.dbs[0] + hb * bp.dbs[1] + (int64_t)row * bp.dbs[2]) * (dp.out_f32 ? 4 : 2);
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            const int kfirst = key0 + t * KT + kb * 32 + 16 * hi;
            if (kfirst >= p.Sk) continue;
            if (dp.out_f32) {
                float* o = reinterpret_cast<float*>(orow) + kfirst;
                if (kfirst + 16 <= p.Sk && bp.dbias_vec) {
#pragma unroll
                    for (int g = 0; g < 4; ++g) *reinterpret_cast<f32x4*>(o + 4 * g) = f32x4{dsum[t][kb][4 * g], dsum[t][kb][4 * g + 1], dsum[t][kb][4 * g + 2], dsum[t][kb][4 * g + 3]};
                } else {
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        if (kfirst + r < p.Sk) o[r] = dsum[t][kb][r];
                }
            } else {
                uint16_t hv[16];
#pragma unroll
                for (int g = 0; g < 2; ++g) {
                    f32x8 x;
#pragma unroll
                    for (int e = 0; e < 8; ++e) x[e] = dsum[t][kb][8 * g + e];
                    const vec8 y = E::cvt8(x);
                    __builtin_memcpy(hv + 8 * g, &y, 16);
                }
                char* o = orow + kfirst * 2;
                if (kfirst + 16 <= p.Sk && bp.dbias_vec) {
                    u32x4 w0, w1;
                    __builtin_memcpy(&w0, hv, 16);
                    __builtin_memcpy(&w1, hv + 8, 16);
                    gstore16(o, w0);
                    gstore16(o + 16, w1);
                } else {
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        if (kfirst + r < p.Sk) reinterpret_cast<uint16_t*>(o)[r] = hv[r];
                }
            }
        }
}

}  // namespace fasn

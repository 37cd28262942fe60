// ubench4.cpp — like ubench3 but with the REAL data dependencies of the attention tile and, optionally, its LDS traffic:
//   phase A: S = mfma(K-frag, Q-frag) x8 (two accumulators), phase B: p = exp2(fma(S)), sum, cvt -> P-frags,
//   phase C: O = mfma(V-frag, P-frag) x8.  mode bit0: K/V fragments come from LDS reads (ds_read_b128 / ds_read_b64_tr_b16).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("hip error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) float f32x8;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;

template <int LDS, int PRIO = 0>
__global__ void __launch_bounds__(512) kern(float* out, int iters) {
    if (PRIO && (threadIdx.x >> 8)) __builtin_amdgcn_s_setprio(PRIO);   // waves 4-7 (second wave of each SIMD)
    __shared__ __attribute__((aligned(16))) char lds[32768];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 8192; i += blockDim.x) ((unsigned*)lds)[i] = 0x3c003c00u + i;
    __syncthreads();
    bf16x8 q[4];
    for (int s = 0; s < 4; ++s) for (int i = 0; i < 8; ++i) q[s][i] = (__bf16)(0.01f * (lane + i + s));
    f32x16 o[2] = {};
    float l = 0.f;
    const float c = 0.18f, m = 0.3f;
    for (int it = 0; it < iters; ++it) {
        f32x16 sa[2] = {};
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                bf16x8 kf;
                if (LDS) { u32x4 raw = *(__attribute__((address_space(3))) u32x4*)(lds + ((kb * 32 + (lane & 31)) * 128 + ((2 * s + (lane >> 5)) ^ ((lane >> 1) & 7)) * 16)); __builtin_memcpy(&kf, &raw, 16); }
                else kf = q[(s + kb) & 3];
                sa[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, q[s], sa[kb], 0, 0, 0);
            }
        bf16x8 pf[2][2];
        float rs = 0.f;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int t2 = 0; t2 < 2; ++t2) {
                f32x8 x;
#pragma unroll
                for (int e = 0; e < 8; ++e) { x[e] = __builtin_amdgcn_exp2f(__builtin_fmaf(sa[kb][8 * t2 + e], c, -m)); rs += x[e]; }
                pf[kb][t2] = __builtin_convertvector(x, bf16x8);
            }
        l += rs;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
                for (int d = 0; d < 2; ++d) {
                    bf16x8 vf;
                    if (LDS) {
                        const int off = 16384 + (kb * 32 + 16 * t2 + 4 * (lane >> 5) + ((lane & 15) >> 2)) * 128 + d * 64 + ((lane >> 4) & 1) * 32 + (lane & 3) * 8;
                        s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(lds + off));
                        s16x4 b2 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(lds + off + 1024));
                        s16x8 ab = __builtin_shufflevector(a, b2, 0, 1, 2, 3, 4, 5, 6, 7);
                        __builtin_memcpy(&vf, &ab, 16);
                    } else vf = q[(d + t2) & 3];
                    o[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf[kb][t2], o[d], 0, 0, 0);
                }
    }
    float s = l;
    for (int d = 0; d < 2; ++d) for (int r = 0; r < 16; ++r) s += o[d][r];
    out[blockIdx.x * 512 + threadIdx.x] = s;
}

template <int LDS, int PRIO = 0>
static float run(int nwaves, int blocks_per_cu, int iters) {
    float* d;
    CHECK(hipMalloc(&d, 256 * 8 * 512 * 4));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL((kern<LDS, PRIO>), dim3(256 * blocks_per_cu), dim3(nwaves * 64), 0, 0, d, 10);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL((kern<LDS, PRIO>), dim3(256 * blocks_per_cu), dim3(nwaves * 64), 0, 0, d, iters);
    CHECK(hipEventRecord(e1, 0));
    CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    CHECK(hipFree(d));
    return ms;
}

int main() {
    const int it = 20000;
    printf("per iteration and wave: 16 MFMA + 32 (fma+exp+add) + 16 cvt_pk, real dependencies; ns per wave-iteration per SIMD\n");
    for (int w : {1, 2, 3, 4}) {
        // w waves per SIMD: w blocks of 4 waves per CU
        float a = run<0>(4, w, it), b = run<1>(4, w, it);
        printf("%d wave(s)/SIMD (independent 4-wave blocks): regs-only %.1f ns   with LDS fragment reads %.1f ns   (MFMA alone = 256 ns)\n", w, a * 1e6 / it / w, b * 1e6 / it / w);
    }
    float a = run<0>(8, 1, it), b = run<1>(8, 1, it);
    printf("2 waves/SIMD as one 8-wave block: regs-only %.1f ns   with LDS %.1f ns\n", a * 1e6 / it / 2, b * 1e6 / it / 2);
    for (int pr = 1; pr <= 3; pr += 2) {
        float c = pr == 1 ? run<0, 1>(8, 1, it) : run<0, 3>(8, 1, it), d = pr == 1 ? run<1, 1>(8, 1, it) : run<1, 3>(8, 1, it);
        printf("  same, waves 4-7 at s_setprio %d: regs-only %.1f ns   with LDS %.1f ns\n", pr, c * 1e6 / it / 2, d * 1e6 / it / 2);
    }
    float e = run<0, 3>(16, 1, it);
    printf("4 waves/SIMD as one 16-wave block, waves 4-15 at prio 3: regs-only %.1f ns\n", e * 1e6 / it / 4);
    return 0;
}

// ubench2.cpp — does an MFMA stream overlap with a VALU/transcendental stream on the SAME SIMD (gfx950)?
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("hip error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;

// role per wave: 0 = MFMA only, 1 = EXP only, 2 = FMA only, 3 = interleave 1 MFMA + K exp, 4 = interleave MFMA + K fma
template <int K>
__global__ void __launch_bounds__(512) kern(float* out, int iters, int role_lo, int role_hi, int nw) {
    const int wave = threadIdx.x >> 6;
    const int role = __builtin_amdgcn_readfirstlane(wave < 4 ? role_lo : role_hi);
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(0.001f * threadIdx.x + i); b[i] = (__bf16)(0.002f * i); }
    f32x16 acc[4] = {};
    float v[32];
    for (int i = 0; i < 32; ++i) v[i] = 0.5f + 0.001f * i;
    float c = 0.999f;
    if (role == 0) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int j = 0; j < 16; ++j) acc[j & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[j & 3], 0, 0, 0);
        }
    } else if (role == 1) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int j = 0; j < 32; ++j) asm volatile("v_exp_f32 %0, %0" : "+v"(v[j]));
        }
    } else if (role == 2) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int j = 0; j < 32; ++j) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[j]) : "v"(c));
        }
    } else if (role == 3) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                acc[j & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[j & 3], 0, 0, 0);
#pragma unroll
                for (int e = 0; e < K; ++e) asm volatile("v_exp_f32 %0, %0" : "+v"(v[(j * K + e) & 31]));
            }
        }
    } else {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                acc[j & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[j & 3], 0, 0, 0);
#pragma unroll
                for (int e = 0; e < K; ++e) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[(j * K + e) & 31]) : "v"(c));
            }
        }
    }
    float s = 0;
    for (int i = 0; i < 32; ++i) s += v[i];
    for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) s += acc[j][r];
    out[blockIdx.x * 512 + threadIdx.x] = s;
}

template <int K>
float run(int nwaves, int role_lo, int role_hi, int iters) {
    float* d;
    CHECK(hipMalloc((void**)&d, 256 * 512 * 4));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(kern<K>, dim3(256), dim3(nwaves * 64), 0, 0, d, 10, role_lo, role_hi, nwaves);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(kern<K>, dim3(256), dim3(nwaves * 64), 0, 0, d, iters, role_lo, role_hi, nwaves);
    CHECK(hipEventRecord(e1, 0));
    CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    CHECK(hipFree(d));
    return ms;
}

int main() {
    const int it = 20000;
    float t;
    t = run<1>(4, 0, 0, it); printf("4 waves MFMA only (1/SIMD)            : %.3f ms  -> %.1f ns per 16 MFMA (%.1f cyc/MFMA @2.4)\n", t, t * 1e6 / it, t * 1e6 / it / 16 * 2.4);
    t = run<1>(8, 0, 0, it); printf("8 waves MFMA only (2/SIMD)            : %.3f ms\n", t);
    t = run<1>(4, 1, 1, it); printf("4 waves EXP only, 32 exp/iter          : %.3f ms  -> %.2f ns per exp\n", t, t * 1e6 / it / 32);
    t = run<1>(8, 1, 1, it); printf("8 waves EXP only                       : %.3f ms\n", t);
    t = run<1>(4, 2, 2, it); printf("4 waves FMA only, 32 fma/iter          : %.3f ms\n", t);
    t = run<1>(8, 2, 2, it); printf("8 waves FMA only                       : %.3f ms\n", t);
    t = run<1>(8, 0, 1, it); printf("8 waves: 4 MFMA + 4 EXP (same SIMDs?)  : %.3f ms   (sum would be MFMA4+EXP4, max = overlap)\n", t);
    t = run<1>(8, 0, 2, it); printf("8 waves: 4 MFMA + 4 FMA                : %.3f ms\n", t);
    t = run<1>(8, 1, 2, it); printf("8 waves: 4 EXP + 4 FMA                 : %.3f ms\n", t);
    t = run<1>(4, 3, 3, it); printf("4 waves interleave MFMA + 1 exp        : %.3f ms\n", t);
    t = run<2>(4, 3, 3, it); printf("4 waves interleave MFMA + 2 exp        : %.3f ms\n", t);
    t = run<4>(4, 3, 3, it); printf("4 waves interleave MFMA + 4 exp        : %.3f ms\n", t);
    t = run<2>(4, 4, 4, it); printf("4 waves interleave MFMA + 2 fma        : %.3f ms\n", t);
    t = run<4>(4, 4, 4, it); printf("4 waves interleave MFMA + 4 fma        : %.3f ms\n", t);
    t = run<8>(4, 4, 4, it); printf("4 waves interleave MFMA + 8 fma        : %.3f ms\n", t);
    t = run<2>(8, 3, 3, it); printf("8 waves interleave MFMA + 2 exp        : %.3f ms\n", t);
    t = run<4>(8, 4, 4, it); printf("8 waves interleave MFMA + 4 fma        : %.3f ms\n", t);
    return 0;
}

// ubench3.cpp — two waves per SIMD, each alternating a 16-MFMA phase and a softmax-like VALU phase (32 fma + 32 exp +
// 32 add + 16 cvt_pk): do the phases of the two co-resident waves overlap, and does a forced anti-phase start help?
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("hip error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;

__device__ __forceinline__ void mfma_phase(f32x16 (&acc)[4], bf16x8 a, bf16x8 b) {
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[j & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[j & 3], 0, 0, 0);
}
__device__ __forceinline__ void valu_phase(float (&v)[32], float c) {
#pragma unroll
    for (int j = 0; j < 32; ++j) {
        asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[j]) : "v"(c));
        asm volatile("v_exp_f32 %0, %0" : "+v"(v[j]));
        asm volatile("v_add_f32 %0, %0, %1" : "+v"(v[j]) : "v"(c));
    }
#pragma unroll
    for (int j = 0; j < 32; j += 2) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(v[j]) : "v"(v[j + 1]));
}

// mode 0: every wave starts with MFMA phase; mode 1: waves 4-7 start with the VALU phase (anti-phase)
__global__ void __launch_bounds__(512) kern(float* out, int iters, int mode, int only) {
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(0.001f * threadIdx.x + i); b[i] = (__bf16)(0.002f * i); }
    f32x16 acc[4] = {};
    float v[32];
    for (int i = 0; i < 32; ++i) v[i] = 0.5f + 0.001f * i;
    const float c = 0.999f;
    if (only == 1) { for (int it = 0; it < iters; ++it) mfma_phase(acc, a, b); }
    else if (only == 2) { for (int it = 0; it < iters; ++it) valu_phase(v, c); }
    else {
        if (mode == 1 && wave >= 4) valu_phase(v, c);
        for (int it = 0; it < iters; ++it) {
            mfma_phase(acc, a, b);
            valu_phase(v, c);
        }
    }
    float s = 0;
    for (int i = 0; i < 32; ++i) s += v[i];
    for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) s += acc[j][r];
    out[blockIdx.x * 512 + threadIdx.x] = s;
}

static float run(int nwaves, int mode, int only, int iters) {
    float* d;
    CHECK(hipMalloc(&d, 256 * 512 * 4));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(kern, dim3(256), dim3(nwaves * 64), 0, 0, d, 10, mode, only);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(kern, dim3(256), dim3(nwaves * 64), 0, 0, d, iters, mode, only);
    CHECK(hipEventRecord(e1, 0));
    CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    CHECK(hipFree(d));
    return ms;
}

int main() {
    const int it = 20000;
    printf("per iteration = 16 MFMA 32x32x16 + (32 fma + 32 exp + 32 add + 16 cvt_pk); times for %d iterations\n", it);
    printf("1 wave/SIMD  MFMA phase only           : %.3f ms\n", run(4, 0, 1, it));
    printf("1 wave/SIMD  VALU phase only           : %.3f ms\n", run(4, 0, 2, it));
    printf("2 waves/SIMD VALU phase only           : %.3f ms\n", run(8, 0, 2, it));
    printf("1 wave/SIMD  alternating               : %.3f ms\n", run(4, 0, 0, it));
    printf("2 waves/SIMD alternating, same start   : %.3f ms   (per wave the same work as the line above)\n", run(8, 0, 0, it));
    printf("2 waves/SIMD alternating, anti-phase   : %.3f ms\n", run(8, 1, 0, it));
    return 0;
}

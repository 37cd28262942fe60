// ubench8.cpp — MFMA 32x32x16 bf16 throughput vs the number of independent accumulator chains per wave (dependent MFMAs
// are NCH instructions apart) and waves per SIMD.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("hip error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
template <int NCH>
__global__ void __launch_bounds__(256, 2) kern(float* out, int iters) {
    const int lane = threadIdx.x & 63;
    bf16x8 a[4], b[4];
    for (int s = 0; s < 4; ++s) for (int i = 0; i < 8; ++i) { a[s][i] = (__bf16)(0.01f * (lane + i + s)); b[s][i] = (__bf16)(0.02f * (lane - i + s)); }
    f32x16 acc[NCH] = {};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 48; ++j) {
            asm volatile("" : "+v"(a[j & 3]));
            acc[j % NCH] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[j & 3], b[(j >> 2) & 3], acc[j % NCH], 0, 0, 0);
        }
    }
    float s = 0.f;
    for (int c = 0; c < NCH; ++c) for (int r = 0; r < 16; ++r) s += acc[c][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NCH>
static void run() {
    float* d;
    CHECK(hipMalloc(&d, 256 * 8 * 256 * 4));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const int it = 5000;
    printf("%d chain(s):", NCH);
    for (int w : {1, 2, 3}) {
        hipLaunchKernelGGL(kern<NCH>, dim3(256 * w), dim3(256), 0, 0, d, 100);
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(kern<NCH>, dim3(256 * w), dim3(256), 0, 0, d, it);
        CHECK(hipEventRecord(e1, 0));
        CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        printf("  %dw %5.2f ns/MFMA", w, ms * 1e6 / it / 48 / w);
    }
    printf("\n");
    CHECK(hipFree(d));
}
int main() {
    printf("v_mfma_f32_32x32x16_bf16, SIMD time per instruction (32 cycles = 13.3 ns at 2.4 GHz, 15.2 ns at 2.1 GHz)\n");
    run<1>(); run<2>(); run<3>(); run<4>(); run<6>(); run<8>();
    return 0;
}

// ubench6.cpp — issue rate of the exponential flavours on gfx950 (one wave per SIMD and two), ns per wave-instruction.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("hip error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
template <int KIND>
__global__ void __launch_bounds__(256) kern(float* out, int iters) {
    float x[16];
    for (int i = 0; i < 16; ++i) x[i] = 0.001f * (threadIdx.x + i);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            if (KIND == 0) asm volatile("v_exp_f32 %0, %0" : "+v"(x[i]));
            if (KIND == 1) asm volatile("v_exp_f16 %0, %0" : "+v"(x[i]));
            if (KIND == 2) asm volatile("v_exp_legacy_f32 %0, %0" : "+v"(x[i]));
            if (KIND == 3) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(x[i]));
            if (KIND == 4) asm volatile("v_pk_fma_f32 %0, %0, %0, %0" : "+v"(*(double*)&x[i & ~1]));
            if (KIND == 5) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %0" : "+v"(x[i]));
            if (KIND == 6) asm volatile("v_rcp_f32 %0, %0" : "+v"(x[i]));
            if (KIND == 7) asm volatile("v_log_f32 %0, %0" : "+v"(x[i]));
            if (KIND == 8) asm volatile("v_pk_mul_f16 %0, %0, %0" : "+v"(x[i]));
            if (KIND == 9) asm volatile("v_sqrt_f32 %0, %0" : "+v"(x[i]));
        }
    }
    float s = 0.f;
    for (int i = 0; i < 16; ++i) s += x[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int KIND>
static void run(const char* name) {
    float* d;
    CHECK(hipMalloc(&d, 256 * 8 * 256 * 4));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const int it = 20000;
    printf("%-22s", name);
    for (int w : {1, 2, 4}) {
        hipLaunchKernelGGL(kern<KIND>, dim3(256 * w), dim3(256), 0, 0, d, 100);
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(kern<KIND>, dim3(256 * w), dim3(256), 0, 0, d, it);
        CHECK(hipEventRecord(e1, 0));
        CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        printf("  %d wave/SIMD: %5.2f ns/instr", w, ms * 1e6 / it / 16 / w);
    }
    printf("\n");
    CHECK(hipFree(d));
}
int main() {
    printf("16 independent chains per wave; ns per wave-instruction per SIMD (4 cycles at 2.4 GHz = 1.67 ns)\n");
    run<3>("v_fma_f32"); run<4>("v_pk_fma_f32"); run<5>("v_cvt_pk_bf16_f32"); run<8>("v_pk_mul_f16");
    run<0>("v_exp_f32"); run<1>("v_exp_f16"); run<2>("v_exp_legacy_f32"); run<6>("v_rcp_f32"); run<7>("v_log_f32"); run<9>("v_sqrt_f32");
    return 0;
}

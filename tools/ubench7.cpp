// ubench7.cpp — is the single-wave VALU rate an issue limit or a latency? N independent chains of bounded v_fma_f32
// (x = x*a + b), one wave per SIMD: if ns/instr falls as N grows, a lone wave was waiting on dependent results.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("hip error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
template <int N, int KIND>
__global__ void __launch_bounds__(256) kern(float* out, int iters, float a, float b) {
    float x[N];
    for (int i = 0; i < N; ++i) x[i] = 0.001f * (threadIdx.x + i);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 64 / N; ++r)
#pragma unroll
            for (int i = 0; i < N; ++i) {
                if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(a), "v"(b));
                if (KIND == 1) asm volatile("v_exp_f32 %0, %0" : "+v"(x[i]));
                if (KIND == 2) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x[i]) : "v"(b));
            }
    }
    float s = 0.f;
    for (int i = 0; i < N; ++i) s += x[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int N, int KIND>
static void run(const char* name) {
    float* d;
    CHECK(hipMalloc(&d, 256 * 8 * 256 * 4));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const int it = 10000;
    printf("%-12s %2d chains:", name, N);
    for (int w : {1, 2, 3, 4}) {
        hipLaunchKernelGGL((kern<N, KIND>), dim3(256 * w), dim3(256), 0, 0, d, 100, 0.5f, 0.25f);
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL((kern<N, KIND>), dim3(256 * w), dim3(256), 0, 0, d, it, 0.5f, 0.25f);
        CHECK(hipEventRecord(e1, 0));
        CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        printf("  %dw %5.2f", w, ms * 1e6 / it / 64 / w);
    }
    printf("   ns per wave-instruction per SIMD\n");
    CHECK(hipFree(d));
}
int main() {
    printf("64 instructions per loop iteration, N independent dependency chains; 1-4 waves per SIMD\n");
    run<1, 0>("v_fma_f32"); run<2, 0>("v_fma_f32"); run<4, 0>("v_fma_f32"); run<8, 0>("v_fma_f32"); run<16, 0>("v_fma_f32"); run<32, 0>("v_fma_f32"); run<64, 0>("v_fma_f32");
    run<1, 1>("v_exp_f32"); run<4, 1>("v_exp_f32"); run<16, 1>("v_exp_f32"); run<64, 1>("v_exp_f32");
    run<1, 2>("v_add_f32"); run<16, 2>("v_add_f32"); run<64, 2>("v_add_f32");
    return 0;
}

// ubench.cpp — per-instruction VALU throughput probes on gfx950 (developer tool).
// Each kernel runs N iterations of 32 independent instances of one instruction per wave; reports cycles per
// wave-instruction per SIMD at 1, 2, 4 waves per SIMD (grid = 256 CUs * waves).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("hip error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
typedef __attribute__((ext_vector_type(2))) float f2;

template <int OP>
__global__ void __launch_bounds__(256) k(float* out, int iters, float seed) {
    float v[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) v[i] = seed + i * 0.001f + threadIdx.x * 1e-6f;
    float c = seed * 0.5f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 32; ++i) {
            if (OP == 0) asm volatile("v_exp_f32 %0, %0" : "+v"(v[i]));
            if (OP == 1) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[i]) : "v"(c));
            if (OP == 2) asm volatile("v_add_f32 %0, %0, %1" : "+v"(v[i]) : "v"(c));
            if (OP == 3) asm volatile("v_max3_f32 %0, %0, %1, %1" : "+v"(v[i]) : "v"(c));
            if (OP == 4) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(v[i]) : "v"(c));
            if (OP == 6) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(v[i]) : "v"(c));
            if (OP == 7) asm volatile("v_ldexp_f32 %0, %0, %1" : "+v"(v[i]) : "v"(c));
            if (OP == 8) asm volatile("v_rndne_f32 %0, %0" : "+v"(v[i]));
            if (OP == 9) asm volatile("v_cvt_i32_f32 %0, %0" : "+v"(v[i]));
            if (OP == 10) asm volatile("v_lshl_add_u32 %0, %0, 23, %1" : "+v"(v[i]) : "v"(c));
            if (OP == 11) asm volatile("v_exp_f32 %0, %1" : "=v"(v[i]) : "v"(c));  // independent of previous value
        }
        if (OP == 5) {
#pragma unroll
            for (int i = 0; i < 32; i += 2) {
                f2 x = {v[i], v[i + 1]};
                f2 cc = {c, c};
                asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(x) : "v"(cc));
                v[i] = x[0]; v[i + 1] = x[1];
            }
        }
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < 32; ++i) s += v[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int OP>
void run(const char* name, int per_iter) {
    float* d;
    CHECK(hipMalloc((void**)&d, 256 * 8 * 256 * 4));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const int iters = 20000;
    for (int wps : {1, 2, 4, 8}) {
        // block = 256 threads = 4 waves = 1 wave per SIMD; wps blocks per CU
        dim3 grid(256 * wps);
        hipLaunchKernelGGL(k<OP>, grid, dim3(256), 0, 0, d, 100, 1.0f);
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(k<OP>, grid, dim3(256), 0, 0, d, iters, 1.0f);
        CHECK(hipEventRecord(e1, 0));
        CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        // per SIMD: wps waves, each iters*per_iter instructions
        double instr = (double)wps * iters * per_iter;
        double ns_per = ms * 1e6 / instr;
        printf("%-22s waves/SIMD=%d  %.3f ns per wave-instr per SIMD  (= %.2f cycles @2.4GHz, %.2f @1.9GHz)\n", name, wps, ns_per, ns_per * 2.4, ns_per * 1.9);
    }
    CHECK(hipFree(d));
}

int main() {
    run<0>("v_exp_f32 (dep chain)", 32);
    run<11>("v_exp_f32 (indep)", 32);
    run<1>("v_fma_f32", 32);
    run<2>("v_add_f32", 32);
    run<6>("v_mul_f32", 32);
    run<3>("v_max3_f32", 32);
    run<4>("v_cvt_pk_bf16_f32", 32);
    run<5>("v_pk_fma_f32", 16);
    run<7>("v_ldexp_f32", 32);
    run<8>("v_rndne_f32", 32);
    run<9>("v_cvt_i32_f32", 32);
    run<10>("v_lshl_add_u32", 32);
    return 0;
}

// ubench_atomic.cpp - how fast can partial dQ tiles be accumulated into one fp32 [BH][Sq][D] buffer in HBM/L2?
// Emulates the traffic of a fused backward: workgroup (bh, key block) walks the q-tiles (32 rows) of its head and adds a
// [32][D] fp32 partial per tile; 8 waves, each owning a 16 x 16 piece (v_mfma_f32_16x16x32 C layout: col = lane&15,
// rows 4*(lane>>4) + r) or a 32-row x 32-col block (32x32x16 C layout). Variants: hardware fp32 atomics (agent / workgroup
// scope), packed bf16 atomics, plain stores, plain read-modify-write.
//   hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics tools/ubench_atomic.cpp -o tools/ubench_atomic
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int VAR, int D, int LAYOUT>
__global__ void __launch_bounds__(512) acc_kernel(float* acc, int nbh, int Sq, int nkb, int spin, int xaware = 1) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // XCD-aware: blocks b, b+8, ... share an XCD; give one XCD whole heads
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    const int bh = xaware ? (j / nkb) * 8 + xcd : (int)(blockIdx.x / nkb), kb = xaware ? j % nkb : (int)(blockIdx.x % nkb);
    if (bh >= nbh) return;
    float* base = acc + (size_t)bh * Sq * D;
    const float val = 1.0f + kb;
    float sink = 0.f;
    for (int t = 0; t < Sq / 32; ++t) {
        for (int s = 0; s < spin; ++s) sink = __builtin_fmaf(sink, 1.0001f, 0.5f);
        if (LAYOUT == 0) {   // 16x16 pieces: 8 waves cover [32][64] (D = 64) - wave: rows 16*(w&1), cols 16*(w>>1)
            static_assert(D == 64 || LAYOUT != 0, "");
            const int col = 16 * (wave >> 1) + (lane & 15);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = t * 32 + 16 * (wave & 1) + 4 * (lane >> 4) + r;
                float* p = base + (size_t)row * D + col;
                if (VAR == 0) atomicAdd(p, val);
                else if (VAR == 1) unsafeAtomicAdd(p, val);
                else if (VAR == 2) __hip_atomic_fetch_add(p, val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                else if (VAR == 4) *p = val;
                else if (VAR == 5) *p += val;
            }
        } else {   // 32x32 blocks (lane&31 = col, rows (r&3) + 8*(r>>2) + 4*hi): D/32 blocks per tile row of 32 q, split over waves
            const int nblk = D / 32;   // 2 or 4 blocks per 32-row tile; waves >= nblk idle
            if (wave < nblk) {
                const int col = 32 * wave + (lane & 31), hi = lane >> 5;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = t * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                    float* p = base + (size_t)row * D + col;
                    if (VAR == 0) atomicAdd(p, val);
                    else if (VAR == 1) unsafeAtomicAdd(p, val);
                    else if (VAR == 2) __hip_atomic_fetch_add(p, val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    else if (VAR == 4) *p = val;
                    else if (VAR == 5) *p += val;
                }
            }
        }
    }
    if (sink == 123.456f) acc[0] = sink;
}

// packed bf16 atomics: [BH][Sq][D] bf16, each lane adds 2 bf16 (one dword); 16x16 piece -> lane holds rows 4g..4g+3 of one
// column, so pairs along d need a lane exchange in the real kernel; here: lane owns (row, 2 cols) directly.
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
template <int D>
__global__ void __launch_bounds__(512) acc_pk_kernel(bf16x2_t* acc, int nbh, int Sq, int nkb) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    const int bh = (j / nkb) * 8 + xcd;
    if (bh >= nbh) return;
    bf16x2_t* base = acc + (size_t)bh * Sq * (D / 2);
    bf16x2_t v = {(__bf16)1.0f, (__bf16)0.5f};
    for (int t = 0; t < Sq / 32; ++t) {
        // 8 waves x 64 lanes x 2 dwords = [32][32 dwords]
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int row = t * 32 + wave * 4 + 2 * r + (lane >> 5);
            __builtin_amdgcn_global_atomic_fadd_v2bf16((__attribute__((address_space(1))) bf16x2_t*)(base + (size_t)row * (D / 2) + (lane & 31)), v);
        }
    }
}

template <typename F>
float time_it(F f, int iters) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    f();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    for (int i = 0; i < iters; ++i) f();
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms;
    CK(hipEventElapsedTime(&ms, a, b));
    return ms / iters;
}

int main(int argc, char** argv) {
    const int nbh = 128, Sq = 4096, D = 64;
    const size_t n = (size_t)nbh * Sq * D;
    float* acc;
    CK(hipMalloc(&acc, n * 4));
    CK(hipMemset(acc, 0, n * 4));
    const char* names[] = {"atomicAdd", "unsafeAtomicAdd", "fetch_add wg-scope", "-", "plain store", "plain rmw"};
    for (int nkb : {8, 16}) {
        const int grid = nbh * nkb;
        const double bytes = (double)n * 4 * nkb;
        printf("== %d key blocks per head, grid %d x 512 threads, %.2f GB of partials per launch\n", nkb, grid, bytes / 1e9);
#define RUN(VAR, LAYOUT, SPIN)                                                                                              \
    {                                                                                                                        \
        float ms = time_it([&] { acc_kernel<VAR, 64, LAYOUT><<<grid, 512>>>(acc, nbh, Sq, nkb, SPIN); }, 5);               \
        printf("  %-20s layout %s spin %4d : %8.3f ms  %7.1f GB/s\n", names[VAR], LAYOUT ? "32x32" : "16x16", SPIN, ms, bytes / ms / 1e6); \
    }
        RUN(0, 0, 0) RUN(1, 0, 0) RUN(2, 0, 0) RUN(4, 0, 0) RUN(5, 0, 0)
        RUN(0, 1, 0) RUN(1, 1, 0) RUN(2, 1, 0) RUN(4, 1, 0)
        RUN(1, 0, 300) RUN(1, 0, 1000) RUN(4, 0, 300) RUN(4, 0, 1000)
        {
            float ms = time_it([&] { acc_pk_kernel<64><<<grid, 512>>>((bf16x2_t*)acc, nbh, Sq, nkb); }, 5);
            printf("  %-20s                        : %8.3f ms  %7.1f GB/s (bf16 bytes)\n", "pk_add_bf16", ms, bytes / 2 / ms / 1e6);
        }
    }
    // correctness of the hardware fp32 atomic: every element must equal sum_{kb} (1 + kb)
    CK(hipMemset(acc, 0, n * 4));
    acc_kernel<1, 64, 0><<<nbh * 8, 512>>>(acc, nbh, Sq, 8, 0);
    CK(hipDeviceSynchronize());
    std::vector<float> h(1 << 20);
    CK(hipMemcpy(h.data(), acc + (n - h.size()), h.size() * 4, hipMemcpyDeviceToHost));
    size_t bad = 0;
    for (float x : h) bad += (x != 36.0f);
    printf("unsafeAtomicAdd sum check: %zu wrong of %zu (expect 36)\n", bad, h.size());
    // the same with the key blocks of one head spread over all XCDs (block b -> XCD b % 8): are the L2-side atomics coherent?
    CK(hipMemset(acc, 0, n * 4));
    acc_kernel<1, 64, 0><<<nbh * 8, 512>>>(acc, nbh, Sq, 8, 0, 0);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(h.data(), acc + (n - h.size()), h.size() * 4, hipMemcpyDeviceToHost));
    bad = 0;
    for (float x : h) bad += (x != 36.0f);
    printf("cross-XCD sum check: %zu wrong of %zu (expect 36)\n", bad, h.size());
    {
        float ms = time_it([&] { acc_kernel<1, 64, 0><<<nbh * 8, 512>>>(acc, nbh, Sq, 8, 0, 0); }, 5);
        printf("  unsafeAtomicAdd, heads spread over XCDs: %.3f ms %.1f GB/s\n", ms, (double)n * 4 * 8 / ms / 1e6);
    }
    // zero-fill and convert costs
    {
        float ms = time_it([&] { CK(hipMemsetAsync(acc, 0, n * 4, 0)); }, 5);
        printf("memset %.1f MB: %.3f ms\n", n * 4 / 1e6, ms);
    }
    return 0;
}

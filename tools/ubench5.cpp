// ubench5.cpp — the hand-ordered software-pipelined attention tile (fasn_fwd_pipe.h, BURST 3), registers only, with
// pieces switched off one at a time: which instruction class keeps the MFMA pipe from running under the softmax VALU?
//   per iteration and wave: 16 MFMA 32x32x16 (8 "PV" on 2 accumulators, 8 "QK^T" on 2 accumulators, zero-initialised)
//   and, per MFMA, the softmax of 2 elements per lane of the PREVIOUS QK^T result: pk_fma | 2 exp | add | cvt_pk.
// FLAGS bit0 exp, bit1 pk_fma, bit2 add, bit3 cvt, bit4 MFMA, bit5 VALU input = MFMA output (else loop-invariant registers),
//       bit6 cvt results feed the next iteration's PV MFMAs (else loop-invariant B operand)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <type_traits>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("hip error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;

template <int FLAGS, int ORDER = 0>
__global__ void __launch_bounds__(256, 2) kern(float* out, int iters) {
    constexpr bool EXP = FLAGS & 1, FMA = FLAGS & 2, ADD = FLAGS & 4, CVT = FLAGS & 8, MM = FLAGS & 16, DEP = FLAGS & 32, PDEP = FLAGS & 64;
    const int lane = threadIdx.x & 63;
    bf16x8 q[4], fr[4];
    for (int s = 0; s < 4; ++s) for (int i = 0; i < 8; ++i) { q[s][i] = (__bf16)(0.01f * (lane + i + s)); fr[s][i] = (__bf16)(0.02f * (lane + i - s)); }
    f32x16 o[2] = {}, sacc[2][2], cst[2];
    bf16x8 pf[2][2][2];
    for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) { for (int r = 0; r < 16; ++r) { sacc[a][b][r] = 0.01f * (r + lane + a); cst[b][r] = 0.02f * (r + lane); }
        for (int t = 0; t < 2; ++t) pf[a][b][t] = q[(a + b + t) & 3]; }
    const f32x2 c2 = {0.18f, 0.18f}, m2 = {-0.3f, -0.3f};
    f32x2 rs2 = {0.f, 0.f};
    auto body = [&](auto CSET) {
        constexpr int C = decltype(CSET)::value;
        if (!DEP) { asm volatile("" : "+v"(cst[0])); asm volatile("" : "+v"(cst[1])); }
        f32x2 tq[2] = {{0.1f, 0.2f}, {0.3f, 0.4f}}, xq[2] = {{0.1f, 0.2f}, {0.3f, 0.4f}};
        auto mm = [&](int j) {
            if (!MM) return;
            asm volatile("" : "+v"(fr[j & 3]));   // opaque: nothing is loop-invariant
            if (j < 8) {
                o[j & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr[j & 3], PDEP ? pf[C ^ 1][j >> 2][(j >> 1) & 1] : q[j & 3], o[j & 1], 0, 0, 0);
            } else {
                const int kb = (j - 8) & 1, ks = (j - 8) >> 1;
                f32x16 z; for (int r = 0; r < 16; ++r) z[r] = 0.f;
                sacc[C ^ 1][kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr[j & 3], q[ks], ks == 0 ? z : sacc[C ^ 1][kb], 0, 0, 0);
            }
        };
        auto v_fma = [&](int i) {
            if (i < 0 || i > 15) return;
            const f32x16& src = DEP ? sacc[C][i >> 3] : cst[i >> 3];
            const f32x2 s2 = {src[2 * (i & 7)], src[2 * (i & 7) + 1]};
            if (FMA) tq[i & 1] = __builtin_elementwise_fma(s2, c2, m2); else tq[i & 1] = s2;
        };
        auto v_exp = [&](int i) {
            if (i < 0 || i > 15) return;
            if (EXP) xq[i & 1] = f32x2{__builtin_amdgcn_exp2f(tq[i & 1][0]), __builtin_amdgcn_exp2f(tq[i & 1][1])}; else xq[i & 1] = tq[i & 1];
        };
        auto v_out = [&](int i) {
            if (i < 0 || i > 15) return;
            if (ADD) rs2 += xq[i & 1];
            if (CVT) {
                const bf16x2 h2 = __builtin_convertvector(xq[i & 1], bf16x2);
                const int kb = i >> 3, t2 = (i >> 2) & 1, e = 2 * (i & 3);
                pf[C][kb][t2][e] = h2[0];
                pf[C][kb][t2][e + 1] = h2[1];
            } else if (!ADD) { asm volatile("" ::"v"(xq[i & 1])); }
        };
        constexpr int LEAD = 2;
        if (ORDER == 0) {
#pragma unroll
            for (int i = -LEAD; i < 0; ++i) { v_out(i + LEAD - 2); v_exp(i + LEAD - 1); v_fma(i + LEAD); __builtin_amdgcn_sched_barrier(0); }
#pragma unroll
            for (int i = 0; i < 16; ++i) { mm(i); v_out(i + LEAD - 2); v_exp(i + LEAD - 1); v_fma(i + LEAD); __builtin_amdgcn_sched_barrier(0); }
        } else if (ORDER == 1) {   // exponentials batched: even groups 4 exps, odd groups the packed ops of two pairs
            v_fma(0); v_fma(1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int g = 0; g < 16; ++g) {
                mm(g);
                if ((g & 1) == 0) { v_exp(g); __builtin_amdgcn_sched_barrier(0); v_exp(g + 1); }
                else { v_out(g - 1); v_out(g); v_fma(g + 1); v_fma(g + 2); }
                __builtin_amdgcn_sched_barrier(0);
            }
        } else if (ORDER == 3) {   // deeper skew: exponential two groups after its fma, consumers two groups after the exponential
            f32x2 tq4[4], xq4[4];
            auto fma4 = [&](int i) { if (i < 0 || i > 15) return; const f32x16& src = DEP ? sacc[C][i >> 3] : cst[i >> 3];
                const f32x2 s2 = {src[2 * (i & 7)], src[2 * (i & 7) + 1]}; tq4[i & 3] = __builtin_elementwise_fma(s2, c2, m2); };
            auto exp4 = [&](int i) { if (i < 0 || i > 15) return; xq4[i & 3] = f32x2{__builtin_amdgcn_exp2f(tq4[i & 3][0]), __builtin_amdgcn_exp2f(tq4[i & 3][1])}; };
            auto out4 = [&](int i) { if (i < 0 || i > 15) return; rs2 += xq4[i & 3]; const bf16x2 h2 = __builtin_convertvector(xq4[i & 3], bf16x2);
                const int kb = i >> 3, t2 = (i >> 2) & 1, e = 2 * (i & 3); pf[C][kb][t2][e] = h2[0]; pf[C][kb][t2][e + 1] = h2[1]; };
#pragma unroll
            for (int i = -4; i < 0; ++i) { out4(i); exp4(i + 2); fma4(i + 4); __builtin_amdgcn_sched_barrier(0); }
#pragma unroll
            for (int i = 0; i < 16; ++i) { mm(i); out4(i); exp4(i + 2); fma4(i + 4); __builtin_amdgcn_sched_barrier(0); }
        } else if (ORDER == 4) {   // VALU first, MFMA last in each group
#pragma unroll
            for (int i = -LEAD; i < 0; ++i) { v_out(i + LEAD - 2); v_exp(i + LEAD - 1); v_fma(i + LEAD); __builtin_amdgcn_sched_barrier(0); }
#pragma unroll
            for (int i = 0; i < 16; ++i) { v_out(i + LEAD - 2); v_exp(i + LEAD - 1); v_fma(i + LEAD); mm(i); __builtin_amdgcn_sched_barrier(0); }
        } else if (ORDER == 5) {   // no pinning at all: the compiler's own order
#pragma unroll
            for (int i = 0; i < 16; ++i) mm(i);
#pragma unroll
            for (int i = 0; i < 16; ++i) { v_fma(i); v_exp(i); v_out(i); }
        } else if (ORDER == 6) {   // two MFMAs, then the arithmetic of two pairs
            v_fma(0); v_fma(1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int g = 0; g < 16; g += 2) {
                mm(g); mm(g + 1);
                v_exp(g); v_exp(g + 1); v_out(g - 2); v_out(g - 1); v_fma(g + 2); v_fma(g + 3);
                __builtin_amdgcn_sched_barrier(0);
            }
            v_out(14); v_out(15);
        } else if (ORDER == 7) {   // D = 128 density: 32 MFMAs per 16 pairs (two passes over the 16 MFMAs, softmax every other group)
#pragma unroll
            for (int i = -LEAD; i < 0; ++i) { v_out(i + LEAD - 2); v_exp(i + LEAD - 1); v_fma(i + LEAD); __builtin_amdgcn_sched_barrier(0); }
#pragma unroll
            for (int g = 0; g < 32; ++g) {
                mm(g & 15);
                if ((g & 1) == 0) { const int i = g >> 1; v_out(i + LEAD - 2); v_exp(i + LEAD - 1); v_fma(i + LEAD); }
                __builtin_amdgcn_sched_barrier(0);
            }
        } else if (ORDER == 2) {   // batched by four: groups 4k, 4k+1: 4 exps each; groups 4k+2, 4k+3: packed ops of 4 pairs
            // pairs 4k..4k+3 exponentiated in groups 4k (pairs 4k,4k+1) and 4k+1 (4k+2,4k+3); out in 4k+2 / 4k+3; fma for the next four in 4k+2 / 4k+3
        }
    };
    for (int it = 0; it < iters; it += 2) {
        body(std::integral_constant<int, 0>{});
        body(std::integral_constant<int, 1>{});
    }
    float s = rs2[0] + rs2[1];
    for (int d = 0; d < 2; ++d) for (int r = 0; r < 16; ++r) s += o[d][r] + sacc[0][d][r] + sacc[1][d][r];
    for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) for (int t = 0; t < 2; ++t) s += (float)pf[a][b][t][lane & 7];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int FLAGS, int ORDER = 0>
static void run(const char* what, int it) {
    float* d;
    CHECK(hipMalloc(&d, 256 * 8 * 256 * 4));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    printf("%-58s", what);
    for (int w : {1, 2, 3}) {
        hipLaunchKernelGGL((kern<FLAGS, ORDER>), dim3(256 * w), dim3(256), 0, 0, d, 10);
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL((kern<FLAGS, ORDER>), dim3(256 * w), dim3(256), 0, 0, d, it);
        CHECK(hipEventRecord(e1, 0));
        CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        printf("  %dw %6.1f", w, ms * 1e6 / it / w);
    }
    printf("\n");
    CHECK(hipFree(d));
}

int main() {
    const int it = 20000;
    printf("hand-ordered pipelined tile, registers only; 16 MFMA alone = 256 ns at 2.0 GHz\n");
    run<16>("MFMA only", it);
    run<127>("everything (real dependencies)", it);
    run<127, 1>("everything, exponentials batched 4 per other group", it);
    run<127, 3>("everything, deeper skew (2 groups per stage)", it);
    run<127, 4>("everything, VALU first / MFMA last in a group", it);
    run<127, 5>("everything, compiler's order (MFMA burst, then VALU)", it);
    run<127, 6>("everything, groups of two MFMAs", it);
    run<127, 7>("D=128 density: 32 MFMA + 32 elements (MFMA alone 2x)", it);
    run<16, 7>("D=128 density: the 32 MFMAs alone", it);
    run<16 + 1 + 32, 1>("MFMA + exp only, batched", it);
    run<127 - 64>("everything, PV operand loop-invariant", it);
    run<127 - 32>("everything, VALU input loop-invariant", it);
    run<127 - 32 - 64>("everything, no MFMA<->VALU register dependencies", it);
    run<127 - 1>("no exp", it);
    run<127 - 2>("no pk_fma", it);
    run<127 - 4>("no add", it);
    run<127 - 8 - 64>("no cvt", it);
    run<16 + 1 + 32>("MFMA + exp only", it);
    run<16 + 2 + 32>("MFMA + pk_fma only", it);
    run<16 + 4 + 32>("MFMA + add only", it);
    run<16 + 8 + 32 + 64>("MFMA + cvt only", it);
    return 0;
}

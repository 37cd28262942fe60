// fasn_moments.hip — one pass over a [rows, cols] matrix: per row the raw power sums  sum x, sum x^2, sum x^3, sum x^4
// (fp64 accumulation), from which the host forms mean / variance / skewness / kurtosis. Replaces the 4-6 full passes of
// flash_attention_softmax_n/analysis/statistics.py:9-79 (mean, subtract, pow, mean ... per statistic) for device tensors.
// HBM-bound: every element is read exactly once, 16 bytes per lane per load.
#include <hip/hip_runtime.h>
#include "fasn.h"
#include "fasn_common.h"

namespace fasn {
namespace {

template <int DT> struct Ld;   // 16 bytes -> floats
template <> struct Ld<FASN_DTYPE_F32> {
    static constexpr int N = 4;
    static FASN_DEV void get(const void* p, int64_t i, float (&v)[8]) {
        const f32x4 w = *reinterpret_cast<const f32x4*>(static_cast<const float*>(p) + i);
        for (int e = 0; e < 4; ++e) v[e] = w[e];
    }
    static FASN_DEV float one(const void* p, int64_t i) { return static_cast<const float*>(p)[i]; }
};
template <> struct Ld<FASN_DTYPE_BF16> {
    static constexpr int N = 8;
    static FASN_DEV void get(const void* p, int64_t i, float (&v)[8]) {
        const u32x4 w = *reinterpret_cast<const u32x4*>(static_cast<const uint16_t*>(p) + i);
        for (int e = 0; e < 4; ++e) {
            v[2 * e] = __uint_as_float(w[e] << 16);
            v[2 * e + 1] = __uint_as_float(w[e] & 0xffff0000u);
        }
    }
    static FASN_DEV float one(const void* p, int64_t i) { return ET<bf16_tag>::to_f32(static_cast<const uint16_t*>(p)[i]); }
};
template <> struct Ld<FASN_DTYPE_F16> {
    static constexpr int N = 8;
    static FASN_DEV void get(const void* p, int64_t i, float (&v)[8]) {
        const u32x4 w = *reinterpret_cast<const u32x4*>(static_cast<const uint16_t*>(p) + i);
        for (int e = 0; e < 4; ++e) {
            v[2 * e] = ET<f16_tag>::to_f32((uint16_t)(w[e] & 0xffffu));
            v[2 * e + 1] = ET<f16_tag>::to_f32((uint16_t)(w[e] >> 16));
        }
    }
    static FASN_DEV float one(const void* p, int64_t i) { return ET<f16_tag>::to_f32(static_cast<const uint16_t*>(p)[i]); }
};

// grid = (chunks per row, rows): a row is cut into chunks so that short-and-many and long-and-few both fill the GPU;
// partial sums are added to the output with fp64 atomics (the caller zeroes it).
template <int DT>
__global__ void __launch_bounds__(256) moments_kernel(const void* x, double* out, int64_t cols, int64_t row_stride, int64_t chunk) {
    constexpr int N = Ld<DT>::N;
    const int64_t row = blockIdx.y;
    const int64_t c0 = (int64_t)blockIdx.x * chunk, c1 = min(cols, c0 + chunk);
    const int64_t base = row * row_stride;
    double s1 = 0, s2 = 0, s3 = 0, s4 = 0;
    auto add = [&](float f) {
        const double d = f, d2 = d * d;
        s1 += d;
        s2 += d2;
        s3 += d2 * d;
        s4 += d2 * d2;
    };
    const bool vec = ((base + c0) % N == 0) && (reinterpret_cast<uintptr_t>(x) % 16 == 0);
    if (vec) {
        const int64_t nv = (c1 - c0) / N;
        for (int64_t i = threadIdx.x; i < nv; i += 256) {
            float v[8];
            Ld<DT>::get(x, base + c0 + i * N, v);
#pragma unroll
            for (int e = 0; e < N; ++e) add(v[e]);
        }
        for (int64_t c = c0 + nv * N + threadIdx.x; c < c1; c += 256) add(Ld<DT>::one(x, base + c));
    } else {
        for (int64_t c = c0 + threadIdx.x; c < c1; c += 256) add(Ld<DT>::one(x, base + c));
    }
    __shared__ double red[4][4];
    double s[4] = {s1, s2, s3, s4};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s[k] += __shfl_xor(s[k], o);
    }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0)
        for (int k = 0; k < 4; ++k) red[wave][k] = s[k];
    __syncthreads();
    if (threadIdx.x < 4) {
        const int k = threadIdx.x;
        atomicAdd(out + row * 4 + k, red[0][k] + red[1][k] + red[2][k] + red[3][k]);
    }
}

}  // namespace
}  // namespace fasn

extern "C" int fasn_moments(const void* x, double* sums, int64_t rows, int64_t cols, int64_t row_stride, int32_t dtype,
                            fasn_stream_t stream) {
    using namespace fasn;
    if (x == nullptr || sums == nullptr || rows <= 0 || cols <= 0 || row_stride < cols) return FASN_EINVAL;
    if (rows > 65535) return FASN_EINVAL;   // grid.y
    // about 2048 workgroups in total, at least 4096 elements per chunk
    int64_t chunks = (2048 + rows - 1) / rows;
    int64_t chunk = (cols + chunks - 1) / chunks;
    if (chunk < 4096) chunk = 4096;
    chunk = (chunk + 7) / 8 * 8;
    chunks = (cols + chunk - 1) / chunk;
    const dim3 grid((unsigned)chunks, (unsigned)rows);
    hipStream_t s = (hipStream_t)stream;
    switch (dtype) {
        case FASN_DTYPE_F32: hipLaunchKernelGGL((moments_kernel<FASN_DTYPE_F32>), grid, dim3(256), 0, s, x, sums, cols, row_stride, chunk); break;
        case FASN_DTYPE_BF16: hipLaunchKernelGGL((moments_kernel<FASN_DTYPE_BF16>), grid, dim3(256), 0, s, x, sums, cols, row_stride, chunk); break;
        case FASN_DTYPE_F16: hipLaunchKernelGGL((moments_kernel<FASN_DTYPE_F16>), grid, dim3(256), 0, s, x, sums, cols, row_stride, chunk); break;
        default: return FASN_EDTYPE;
    }
    return hipGetLastError() == hipSuccess ? FASN_OK : FASN_ELAUNCH;
}

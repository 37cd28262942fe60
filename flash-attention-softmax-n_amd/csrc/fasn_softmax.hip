// fasn_softmax.hip — stand-alone softmax_n over the last dimension (fwd + bwd).
//   y_i = exp(x_i - m) / (n * exp(-m) + sum_j exp(x_j - m)),  m = max(x) (and m >= 0 when n > 0 so that
//   n * exp(-m) cannot overflow) — reference: flash_attention_softmax_n/core/functional.py:15-29.
//   dx_i = y_i * (dy_i - sum_j dy_j y_j)   (n enters only through y).
// One workgroup per row at a time, rows taken in a grid-stride loop (the grid is capped: HIP rejects launches of 2^32 or more
// threads, so rows >= 2^24 cannot have a workgroup each); the row is cached in registers when it fits (cols <= 256*EPT), fp32 math.
#include <hip/hip_runtime.h>
#include <math.h>
#include "fasn.h"
#include "fasn_common.h"

namespace fasn {

template <int DT> struct IO;
template <> struct IO<FASN_DTYPE_F32> {
    typedef float T;
    static FASN_DEV float ld(const void* p, int64_t i) { return ((const float*)p)[i]; }
    static FASN_DEV void st(void* p, int64_t i, float v) { ((float*)p)[i] = v; }
};
template <> struct IO<FASN_DTYPE_BF16> {
    static FASN_DEV float ld(const void* p, int64_t i) { return ET<bf16_tag>::to_f32(((const uint16_t*)p)[i]); }
    static FASN_DEV void st(void* p, int64_t i, float v) { ((__bf16*)p)[i] = (__bf16)v; }
};
template <> struct IO<FASN_DTYPE_F16> {
    static FASN_DEV float ld(const void* p, int64_t i) { return (float)((const _Float16*)p)[i]; }
    static FASN_DEV void st(void* p, int64_t i, float v) { ((_Float16*)p)[i] = (_Float16)v; }
};

template <bool IS_MAX>
FASN_DEV float block_reduce(float v, float* red) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float w = __shfl_xor(v, o);
        v = IS_MAX ? fmaxf(v, w) : v + w;
    }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    __syncthreads();
    if (lane == 0) red[wave] = v;
    __syncthreads();
    float r = red[0];
#pragma unroll
    for (int i = 1; i < 4; ++i) r = IS_MAX ? fmaxf(r, red[i]) : r + red[i];
    return r;
}

constexpr int EPT = 16;  // cached elements per thread
constexpr int64_t kMaxRowGrid = 1 << 20;  // workgroups per launch; more rows than that are walked by the grid-stride loop

template <int DT>
__global__ void __launch_bounds__(256) softmax_n_fwd_kernel(const void* x, void* y, int64_t rows, int64_t cols, int64_t xs, int64_t ys, float n) {
    __shared__ float red[4];
    for (int64_t row = blockIdx.x; row < rows; row += gridDim.x) {
    const int64_t xo = row * xs, yo = row * ys;
    const bool cached = cols <= 256 * EPT;
    float v[EPT];
    float mx = -INFINITY;
    if (cached) {
#pragma unroll
        for (int e = 0; e < EPT; ++e) {
            const int64_t c = threadIdx.x + e * 256;
            v[e] = c < cols ? IO<DT>::ld(x, xo + c) : -INFINITY;
            mx = fmaxf(mx, v[e]);
        }
    } else {
        for (int64_t c = threadIdx.x; c < cols; c += 256) mx = fmaxf(mx, IO<DT>::ld(x, xo + c));
    }
    mx = block_reduce<true>(mx, red);
    if (n > 0.f) mx = fmaxf(mx, 0.f);
    if (mx == -INFINITY) mx = 0.f;  // all -inf, n == 0: exp(-inf)/0 -> NaN like the reference
    float sum = 0.f;
    if (cached) {
#pragma unroll
        for (int e = 0; e < EPT; ++e) {
            v[e] = __expf(v[e] - mx);
            sum += v[e];
        }
    } else {
        for (int64_t c = threadIdx.x; c < cols; c += 256) sum += __expf(IO<DT>::ld(x, xo + c) - mx);
    }
    sum = block_reduce<false>(sum, red);
    const float inv = 1.0f / (n * __expf(-mx) + sum);
    if (cached) {
#pragma unroll
        for (int e = 0; e < EPT; ++e) {
            const int64_t c = threadIdx.x + e * 256;
            if (c < cols) IO<DT>::st(y, yo + c, v[e] * inv);
        }
    } else {
        for (int64_t c = threadIdx.x; c < cols; c += 256) IO<DT>::st(y, yo + c, __expf(IO<DT>::ld(x, xo + c) - mx) * inv);
    }
    }
}

template <int DT>
__global__ void __launch_bounds__(256) softmax_n_bwd_kernel(const void* y, const void* dy, void* dx, int64_t rows, int64_t cols, int64_t ys, int64_t dys, int64_t dxs) {
    __shared__ float red[4];
    for (int64_t row = blockIdx.x; row < rows; row += gridDim.x) {
    float dot = 0.f;
    for (int64_t c = threadIdx.x; c < cols; c += 256) dot += IO<DT>::ld(y, row * ys + c) * IO<DT>::ld(dy, row * dys + c);
    dot = block_reduce<false>(dot, red);
    for (int64_t c = threadIdx.x; c < cols; c += 256)
        IO<DT>::st(dx, row * dxs + c, IO<DT>::ld(y, row * ys + c) * (IO<DT>::ld(dy, row * dys + c) - dot));
    }
}

}  // namespace fasn

using namespace fasn;

extern "C" {

int fasn_softmax_n_fwd(const void* x, void* y, int64_t rows, int64_t cols, int64_t x_row_stride, int64_t y_row_stride, float n,
                       int32_t dtype, fasn_stream_t stream) {
    if (x == nullptr || y == nullptr || rows <= 0 || cols <= 0 || !(n >= 0.f)) return FASN_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid((unsigned)(rows < kMaxRowGrid ? rows : kMaxRowGrid));
    switch (dtype) {
        case FASN_DTYPE_F16: hipLaunchKernelGGL(softmax_n_fwd_kernel<FASN_DTYPE_F16>, grid, dim3(256), 0, s, x, y, rows, cols, x_row_stride, y_row_stride, n); break;
        case FASN_DTYPE_BF16: hipLaunchKernelGGL(softmax_n_fwd_kernel<FASN_DTYPE_BF16>, grid, dim3(256), 0, s, x, y, rows, cols, x_row_stride, y_row_stride, n); break;
        case FASN_DTYPE_F32: hipLaunchKernelGGL(softmax_n_fwd_kernel<FASN_DTYPE_F32>, grid, dim3(256), 0, s, x, y, rows, cols, x_row_stride, y_row_stride, n); break;
        default: return FASN_EDTYPE;
    }
    return hipGetLastError() == hipSuccess ? FASN_OK : FASN_ELAUNCH;
}

int fasn_softmax_n_bwd(const void* y, const void* dy, void* dx, int64_t rows, int64_t cols, int64_t y_row_stride, int64_t dy_row_stride,
                       int64_t dx_row_stride, int32_t dtype, fasn_stream_t stream) {
    if (y == nullptr || dy == nullptr || dx == nullptr || rows <= 0 || cols <= 0) return FASN_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid((unsigned)(rows < kMaxRowGrid ? rows : kMaxRowGrid));
    switch (dtype) {
        case FASN_DTYPE_F16: hipLaunchKernelGGL(softmax_n_bwd_kernel<FASN_DTYPE_F16>, grid, dim3(256), 0, s, y, dy, dx, rows, cols, y_row_stride, dy_row_stride, dx_row_stride); break;
        case FASN_DTYPE_BF16: hipLaunchKernelGGL(softmax_n_bwd_kernel<FASN_DTYPE_BF16>, grid, dim3(256), 0, s, y, dy, dx, rows, cols, y_row_stride, dy_row_stride, dx_row_stride); break;
        case FASN_DTYPE_F32: hipLaunchKernelGGL(softmax_n_bwd_kernel<FASN_DTYPE_F32>, grid, dim3(256), 0, s, y, dy, dx, rows, cols, y_row_stride, dy_row_stride, dx_row_stride); break;
        default: return FASN_EDTYPE;
    }
    return hipGetLastError() == hipSuccess ? FASN_OK : FASN_ELAUNCH;
}

}  // extern "C"

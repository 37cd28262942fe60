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

// ---- one WAVE per row, 16-byte loads, no LDS and no barrier: the row lives in registers (NV 16-byte vectors per lane, so rows of
// up to 512 * NV 16-bit or 256 * NV fp32 elements), max and sum by wave shuffles. Needs 16-byte aligned rows and a row length
// that is a multiple of the vector; everything else takes the workgroup-per-row kernels above. One read and one write of every
// element (backward: two reads, one write): HBM-bound.
template <int DT>
struct VecIO {
    static constexpr int EPV = DT == FASN_DTYPE_F32 ? 4 : 8;   // elements per 16-byte vector
    static FASN_DEV void unpack(u32x4 w, float* f) {
        if constexpr (DT == FASN_DTYPE_F32) {
#pragma unroll
            for (int e = 0; e < 4; ++e) f[e] = __uint_as_float(w[e]);
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if constexpr (DT == FASN_DTYPE_BF16) {
                    f[2 * e] = __uint_as_float(w[e] << 16);
                    f[2 * e + 1] = __uint_as_float(w[e] & 0xffff0000u);
                } else {
                    f[2 * e] = ET<f16_tag>::to_f32((uint16_t)(w[e] & 0xffffu));
                    f[2 * e + 1] = ET<f16_tag>::to_f32((uint16_t)(w[e] >> 16));
                }
            }
        }
    }
    static FASN_DEV u32x4 pack(const float* f) {
        u32x4 w;
        if constexpr (DT == FASN_DTYPE_F32) {
#pragma unroll
            for (int e = 0; e < 4; ++e) w[e] = __float_as_uint(f[e]);
        } else if constexpr (DT == FASN_DTYPE_BF16) {
            f32x8 x;
#pragma unroll
            for (int e = 0; e < 8; ++e) x[e] = f[e];
            const bf16x8 y = __builtin_convertvector(x, bf16x8);
            __builtin_memcpy(&w, &y, 16);
        } else {
            f32x8 x;
#pragma unroll
            for (int e = 0; e < 8; ++e) x[e] = f[e];
            const f16x8 y = __builtin_convertvector(x, f16x8);
            __builtin_memcpy(&w, &y, 16);
        }
        return w;
    }
};
FASN_DEV float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}
FASN_DEV float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

template <int DT, int NV>
__global__ void __launch_bounds__(256) softmax_n_fwd_wave_kernel(const char* x, char* y, int64_t rows, int cols, int64_t xs_bytes, int64_t ys_bytes, float n) {
    constexpr int EPV = VecIO<DT>::EPV;
    const int lane = threadIdx.x & 63;
    const int nvec = cols / EPV;   // vectors per row
    for (int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); row < rows; row += (int64_t)gridDim.x * 4) {
        const u32x4* xr = reinterpret_cast<const u32x4*>(x + row * xs_bytes);
        u32x4 raw[NV];
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int vi = lane + 64 * i;
            raw[i] = vi < nvec ? xr[vi] : u32x4{0u, 0u, 0u, 0u};
        }
        float v[NV][EPV];
        float mx = -INFINITY;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            VecIO<DT>::unpack(raw[i], v[i]);
            if (lane + 64 * i < nvec) {
#pragma unroll
                for (int e = 0; e < EPV; ++e) mx = fmaxf(mx, v[i][e]);
            }
        }
        mx = wave_max(mx);
        if (n > 0.f) mx = fmaxf(mx, 0.f);
        if (mx == -INFINITY) mx = 0.f;  // all -inf, n == 0: exp(-inf)/0 -> NaN like the reference
        const float mx2 = mx * kLog2e;
        float sum = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const bool ok = lane + 64 * i < nvec;
#pragma unroll
            for (int e = 0; e < EPV; ++e) {
                v[i][e] = ok ? fast_exp2(__builtin_fmaf(v[i][e], kLog2e, -mx2)) : 0.f;
                sum += v[i][e];
            }
        }
        sum = wave_sum(sum);
        const float inv = 1.0f / (n * __expf(-mx) + sum);
        u32x4* yr = reinterpret_cast<u32x4*>(y + row * ys_bytes);
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int vi = lane + 64 * i;
#pragma unroll
            for (int e = 0; e < EPV; ++e) v[i][e] *= inv;
            if (vi < nvec) yr[vi] = VecIO<DT>::pack(v[i]);
        }
    }
}

template <int DT, int NV>
__global__ void __launch_bounds__(256) softmax_n_bwd_wave_kernel(const char* y, const char* dy, char* dx, int64_t rows, int cols, int64_t ys_bytes, int64_t dys_bytes, int64_t dxs_bytes) {
    constexpr int EPV = VecIO<DT>::EPV;
    const int lane = threadIdx.x & 63;
    const int nvec = cols / EPV;
    for (int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); row < rows; row += (int64_t)gridDim.x * 4) {
        const u32x4* yr = reinterpret_cast<const u32x4*>(y + row * ys_bytes);
        const u32x4* gr = reinterpret_cast<const u32x4*>(dy + row * dys_bytes);
        float yv[NV][EPV], gv[NV][EPV];
        float dot = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int vi = lane + 64 * i;
            const bool ok = vi < nvec;
            VecIO<DT>::unpack(ok ? yr[vi] : u32x4{0u, 0u, 0u, 0u}, yv[i]);
            VecIO<DT>::unpack(ok ? gr[vi] : u32x4{0u, 0u, 0u, 0u}, gv[i]);
#pragma unroll
            for (int e = 0; e < EPV; ++e) dot = __builtin_fmaf(yv[i][e], gv[i][e], dot);
        }
        dot = wave_sum(dot);
        u32x4* xr = reinterpret_cast<u32x4*>(dx + row * dxs_bytes);
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int vi = lane + 64 * i;
#pragma unroll
            for (int e = 0; e < EPV; ++e) yv[i][e] *= gv[i][e] - dot;
            if (vi < nvec) xr[vi] = VecIO<DT>::pack(yv[i]);
        }
    }
}


// ---- one WORKGROUP per row, 16-byte loads, the row in registers (NV vectors per thread: rows of up to 256 * NV vectors = 32768
// 16-bit elements at NV = 16): what the wave kernels do, for rows too long for one wave. One read and one write of every element
// (the element-load kernels above read a long row three times, two bytes per lane: 1.4 TB/s at [4096 x 32768]).
template <int DT, int NV>
__global__ void __launch_bounds__(256) softmax_n_fwd_block_kernel(const char* x, char* y, int64_t rows, int cols, int64_t xs_bytes, int64_t ys_bytes, float n) {
    constexpr int EPV = VecIO<DT>::EPV;
    __shared__ float red[4];
    const int nvec = cols / EPV;
    for (int64_t row = blockIdx.x; row < rows; row += gridDim.x) {
        const u32x4* xr = reinterpret_cast<const u32x4*>(x + row * xs_bytes);
        u32x4 raw[NV];
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int vi = (int)threadIdx.x + 256 * i;
            raw[i] = vi < nvec ? xr[vi] : u32x4{0u, 0u, 0u, 0u};
        }
        float v[NV][EPV];
        float mx = -INFINITY;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            VecIO<DT>::unpack(raw[i], v[i]);
            if ((int)threadIdx.x + 256 * i < nvec) {
#pragma unroll
                for (int e = 0; e < EPV; ++e) mx = fmaxf(mx, v[i][e]);
            }
        }
        mx = block_reduce<true>(mx, red);
        if (n > 0.f) mx = fmaxf(mx, 0.f);
        if (mx == -INFINITY) mx = 0.f;  // all -inf, n == 0: exp(-inf)/0 -> NaN like the reference
        const float mx2 = mx * kLog2e;
        float sum = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const bool ok = (int)threadIdx.x + 256 * i < nvec;
#pragma unroll
            for (int e = 0; e < EPV; ++e) {
                v[i][e] = ok ? fast_exp2(__builtin_fmaf(v[i][e], kLog2e, -mx2)) : 0.f;
                sum += v[i][e];
            }
        }
        sum = block_reduce<false>(sum, red);
        const float inv = 1.0f / (n * __expf(-mx) + sum);
        u32x4* yr = reinterpret_cast<u32x4*>(y + row * ys_bytes);
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int vi = (int)threadIdx.x + 256 * i;
#pragma unroll
            for (int e = 0; e < EPV; ++e) v[i][e] *= inv;
            if (vi < nvec) yr[vi] = VecIO<DT>::pack(v[i]);
        }
    }
}

template <int DT, int NV>
__global__ void __launch_bounds__(256) softmax_n_bwd_block_kernel(const char* y, const char* dy, char* dx, int64_t rows, int cols, int64_t ys_bytes, int64_t dys_bytes, int64_t dxs_bytes) {
    constexpr int EPV = VecIO<DT>::EPV;
    __shared__ float red[4];
    const int nvec = cols / EPV;
    for (int64_t row = blockIdx.x; row < rows; row += gridDim.x) {
        const u32x4* yr = reinterpret_cast<const u32x4*>(y + row * ys_bytes);
        const u32x4* gr = reinterpret_cast<const u32x4*>(dy + row * dys_bytes);
        u32x4 yraw[NV], graw[NV];   // kept packed: unpacked once for the dot product and once for the result
        float dot = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int vi = (int)threadIdx.x + 256 * i;
            const bool ok = vi < nvec;
            yraw[i] = ok ? yr[vi] : u32x4{0u, 0u, 0u, 0u};
            graw[i] = ok ? gr[vi] : u32x4{0u, 0u, 0u, 0u};
            float yv[EPV], gv[EPV];
            VecIO<DT>::unpack(yraw[i], yv);
            VecIO<DT>::unpack(graw[i], gv);
#pragma unroll
            for (int e = 0; e < EPV; ++e) dot = __builtin_fmaf(yv[e], gv[e], dot);
        }
        dot = block_reduce<false>(dot, red);
        u32x4* xr = reinterpret_cast<u32x4*>(dx + row * dxs_bytes);
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int vi = (int)threadIdx.x + 256 * i;
            float yv[EPV], gv[EPV];
            VecIO<DT>::unpack(yraw[i], yv);
            VecIO<DT>::unpack(graw[i], gv);
#pragma unroll
            for (int e = 0; e < EPV; ++e) yv[e] *= gv[e] - dot;
            if (vi < nvec) xr[vi] = VecIO<DT>::pack(yv);
        }
    }
}

template <int DT>
static bool launch_fwd_wave(const void* x, void* y, int64_t rows, int64_t cols, int64_t xs, int64_t ys, float n, hipStream_t s) {
    constexpr int esz = DT == FASN_DTYPE_F32 ? 4 : 2, EPV = VecIO<DT>::EPV;
    if (cols % EPV || (xs * esz) % 16 || (ys * esz) % 16 || reinterpret_cast<uintptr_t>(x) % 16 || reinterpret_cast<uintptr_t>(y) % 16) return false;
    const int64_t nvec = cols / EPV;
    if (nvec > 256 * 16) return false;
    const char* xp = (const char*)x;
    char* yp = (char*)y;
    if (nvec > 64 * 16) {   // too long for one wave's registers: one workgroup per row
        const dim3 bgrid((unsigned)(rows < kMaxRowGrid ? rows : kMaxRowGrid));
#define FASN_SM_FWDB(NV) hipLaunchKernelGGL((softmax_n_fwd_block_kernel<DT, NV>), bgrid, dim3(256), 0, s, xp, yp, rows, (int)cols, xs * esz, ys * esz, n)
        if (nvec <= 256 * 8) FASN_SM_FWDB(8);
        else FASN_SM_FWDB(16);
#undef FASN_SM_FWDB
        return true;
    }
    const int64_t blocks = (rows + 3) / 4;
    const dim3 grid((unsigned)(blocks < kMaxRowGrid ? blocks : kMaxRowGrid));
#define FASN_SM_FWD(NV) hipLaunchKernelGGL((softmax_n_fwd_wave_kernel<DT, NV>), grid, dim3(256), 0, s, xp, yp, rows, (int)cols, xs * esz, ys * esz, n)
    if (nvec <= 64 * 2) FASN_SM_FWD(2);
    else if (nvec <= 64 * 4) FASN_SM_FWD(4);
    else if (nvec <= 64 * 8) FASN_SM_FWD(8);
    else FASN_SM_FWD(16);
#undef FASN_SM_FWD
    return true;
}
template <int DT>
static bool launch_bwd_wave(const void* y, const void* dy, void* dx, int64_t rows, int64_t cols, int64_t ys, int64_t dys, int64_t dxs, hipStream_t s) {
    constexpr int esz = DT == FASN_DTYPE_F32 ? 4 : 2, EPV = VecIO<DT>::EPV;
    if (cols % EPV || (ys * esz) % 16 || (dys * esz) % 16 || (dxs * esz) % 16 || reinterpret_cast<uintptr_t>(y) % 16 || reinterpret_cast<uintptr_t>(dy) % 16 ||
        reinterpret_cast<uintptr_t>(dx) % 16)
        return false;
    const int64_t nvec = cols / EPV;
    if (nvec > 256 * 16) return false;
    const char *yp = (const char*)y, *gp = (const char*)dy;
    char* xp = (char*)dx;
    if (nvec > 64 * 8) {   // too long for one wave's registers: one workgroup per row
        const dim3 bgrid((unsigned)(rows < kMaxRowGrid ? rows : kMaxRowGrid));
#define FASN_SM_BWDB(NV) hipLaunchKernelGGL((softmax_n_bwd_block_kernel<DT, NV>), bgrid, dim3(256), 0, s, yp, gp, xp, rows, (int)cols, ys * esz, dys * esz, dxs * esz)
        if (nvec <= 256 * 4) FASN_SM_BWDB(4);
        else if (nvec <= 256 * 8) FASN_SM_BWDB(8);
        else FASN_SM_BWDB(16);
#undef FASN_SM_BWDB
        return true;
    }
    const int64_t blocks = (rows + 3) / 4;
    const dim3 grid((unsigned)(blocks < kMaxRowGrid ? blocks : kMaxRowGrid));
#define FASN_SM_BWD(NV) hipLaunchKernelGGL((softmax_n_bwd_wave_kernel<DT, NV>), grid, dim3(256), 0, s, yp, gp, xp, rows, (int)cols, ys * esz, dys * esz, dxs * esz)
    if (nvec <= 64 * 2) FASN_SM_BWD(2);
    else if (nvec <= 64 * 4) FASN_SM_BWD(4);
    else FASN_SM_BWD(8);
#undef FASN_SM_BWD
    return true;
}

}  // namespace fasn

using namespace fasn;

extern "C" {

int fasn_softmax_n_fwd(const void* x, void* y, int64_t rows, int64_t cols, int64_t x_row_stride, int64_t y_row_stride, float n,
                       int32_t dtype, fasn_stream_t stream) {
    if (x == nullptr || y == nullptr || rows <= 0 || cols <= 0 || !(n >= 0.f)) return FASN_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    {   // rows that fit a wave's registers and move in 16-byte pieces: one wave per row, no LDS
        bool done = false;
        switch (dtype) {
            case FASN_DTYPE_F16: done = launch_fwd_wave<FASN_DTYPE_F16>(x, y, rows, cols, x_row_stride, y_row_stride, n, s); break;
            case FASN_DTYPE_BF16: done = launch_fwd_wave<FASN_DTYPE_BF16>(x, y, rows, cols, x_row_stride, y_row_stride, n, s); break;
            case FASN_DTYPE_F32: done = launch_fwd_wave<FASN_DTYPE_F32>(x, y, rows, cols, x_row_stride, y_row_stride, n, s); break;
            default: return FASN_EDTYPE;
        }
        if (done) return hipGetLastError() == hipSuccess ? FASN_OK : FASN_ELAUNCH;
    }
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
    {
        bool done = false;
        switch (dtype) {
            case FASN_DTYPE_F16: done = launch_bwd_wave<FASN_DTYPE_F16>(y, dy, dx, rows, cols, y_row_stride, dy_row_stride, dx_row_stride, s); break;
            case FASN_DTYPE_BF16: done = launch_bwd_wave<FASN_DTYPE_BF16>(y, dy, dx, rows, cols, y_row_stride, dy_row_stride, dx_row_stride, s); break;
            case FASN_DTYPE_F32: done = launch_bwd_wave<FASN_DTYPE_F32>(y, dy, dx, rows, cols, y_row_stride, dy_row_stride, dx_row_stride, s); break;
            default: return FASN_EDTYPE;
        }
        if (done) return hipGetLastError() == hipSuccess ? FASN_OK : FASN_ELAUNCH;
    }
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

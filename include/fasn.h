/*
 * fasn.h — C ABI of libfasn: fused attention with softmax_n for AMD MI355X (gfx950 / CDNA4).
 *
 *   softmax_n(x)_i = exp(x_i) / (n + sum_j exp(x_j)),   real n >= 0
 *   O = softmax_n(scale * Q K^T + bias  [masked / causal -> -inf]) V
 *
 * This is the drop-in boundary for the reference's kernel-launch sites:
 *   - flash_attention_softmax_n/core/flash_attn.py:115-124   (torch SDPA call on zero-row-padded K/V)
 *   - flash_attention_softmax_n/core/flash_attn_triton.py:278-291  (_fwd_kernel launch)
 *   - flash_attention_softmax_n/core/flash_attn_triton.py:316-335  (_bwd_preprocess + _bwd_kernel launches)
 * Host code (Python, PyTorch-ROCm) normalises arguments and calls these entry points through ctypes;
 * see INTEGRATION.md for the reference-side binding.
 *
 * Contract
 *   - plain C, no torch types: device pointers, element strides, sizes.
 *   - the library never allocates, frees or synchronises; every buffer (incl. workspace) is caller-owned.
 *   - launches are asynchronous on the hipStream_t passed in (0 = default stream).
 *   - stateless and re-entrant; the current HIP device must be the one owning the pointers.
 *   - returns FASN_OK (0) or a negative FASN_E* code; never throws, never aborts.
 *
 * Tensor layout: 4-D (batch, head, seq, feature) addressed by element strides; the feature stride
 * must be 1 and base pointers / other strides must keep every row 16-byte aligned (strides % 8 elements for the 16-bit
 * types, % 4 for fp32). A stride of 0 is
 * a broadcast dimension (mask, bias, and the head dimension of K/V for shared-KV layouts).
 */
#ifndef FASN_H_
#define FASN_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FASN_ABI_VERSION 6

/* error codes */
#define FASN_OK 0
#define FASN_EINVAL (-1)      /* NULL pointer / non-positive size / bad enum */
#define FASN_EDTYPE (-2)      /* unsupported element type */
#define FASN_EHEADDIM (-3)    /* unsupported head dimension (supported: 32, 64, 128 and - fp16 / bf16 - 256; D == Dv) */
#define FASN_EALIGN (-4)      /* pointer or stride breaks the 16-byte row alignment rule */
#define FASN_ESTRIDE (-5)     /* feature stride != 1 */
#define FASN_ELAUNCH (-6)     /* hipLaunchKernel / hipGetLastError failure */
#define FASN_EUNSUPPORTED (-7)/* valid request this build does not implement (e.g. a reduced bias gradient with fp32 q/k/v or dropout) */
#define FASN_EWORKSPACE (-8)  /* workspace missing or too small */

/* element types of q/k/v/o/do/dq/dk/dv (FASN_DTYPE_F32 = 2, defined below: exact-fp32 MFMA kernels; with fp32 q/k/v the
   bias, if any, is fp32 too and mask / bias / dropout take an element-load instantiation) */
#define FASN_DTYPE_F16 0
#define FASN_DTYPE_BF16 1

/* element type of the additive bias */
#define FASN_BIAS_NONE 0
#define FASN_BIAS_SAME 1 /* same dtype as q */
#define FASN_BIAS_F32 2

typedef void* fasn_stream_t; /* hipStream_t */

/* One 4-D tensor view: ptr + element strides for (batch, head, row, col). */
typedef struct fasn_view4 {
    void* ptr;
    int64_t stride[4];
} fasn_view4;

/*
 * Forward. Replaces: _fwd_kernel launch (flash_attn_triton.py:278-291) and the SDPA call
 * (flash_attn.py:117-124) including its K/V zero-row padding (:66-73) and dense mask
 * materialisation (:87-113), which are expressed here as `softmax_n`, `causal`, `mask`, `bias`.
 */
typedef struct fasn_fwd_args {
    fasn_view4 q;   /* [B,H,Sq,D]  */
    fasn_view4 k;   /* [B,H,Sk,D]  */
    fasn_view4 v;   /* [B,H,Sk,Dv] */
    fasn_view4 o;   /* [B,H,Sq,Dv] out */
    float* lse;     /* [B,H,Sq] fp32 contiguous, out: log(n + sum_j exp(x_ij)) (natural log); may be NULL */
    fasn_view4 mask;/* optional, uint8/bool, nonzero = attend; strides may be 0; ptr NULL = none */
    fasn_view4 bias;/* optional additive bias (added after scaling); ptr NULL = none */
    int32_t bias_dtype; /* FASN_BIAS_* */
    int32_t dtype;      /* FASN_DTYPE_* */
    int32_t B, H, Sq, Sk, D, Dv;
    float scale;        /* multiplies q.k before bias (reference default 1/sqrt(D)) */
    float softmax_n;    /* n >= 0, real-valued */
    int32_t causal;     /* bottom-right aligned: key j visible to row i iff j <= i + Sk - Sq */
    float dropout_p;    /* in [0,1): attention-weight dropout; realised as thr/65536 with thr = round(65536 p) in [1,65535] */
    uint64_t seed, offset; /* dropout stream: the keep bit of (b,h,row,key) is a pure function of (seed, offset, indices);
                              pass the SAME values to fasn_bwd (see flash-attention-softmax-n_amd/dropout.py) */
    int32_t kv_group;      /* grouped-query attention (ABI 2): query head h reads K/V head h / kv_group, i.e. k and v are
                              [B, H / kv_group, Sk, D] addressed through their head stride; 0 or 1 = one K/V head per query
                              head. fasn_bwd writes dk / dv per K/V head, [B, H / kv_group, Sk, D]: each dK/dV workgroup walks the
                              query heads of its group and accumulates their contributions in registers (fp32). */
    const uint64_t* rng_state; /* optional (ABI 3): DEVICE pointer to {seed, offset} (8-byte aligned). When set, the kernels read the
                              dropout stream position from device memory instead of `seed` / `offset` above, so a captured HIP graph
                              that also captures fasn_rng_advance() draws a fresh mask on every replay. Pass the same pointer (and the
                              same contents) to fasn_bwd. NULL = use the by-value fields. */
} fasn_fwd_args;

/*
 * Backward. Replaces _bwd_preprocess + _bwd_kernel (flash_attn_triton.py:316-335), with the
 * softmax_n-correct LSE (the reference's Triton backward drops n).
 * dq/dk/dv are written (not accumulated) in `dtype`. `delta` is a [B,H,Sq] fp32 scratch the
 * caller provides.
 *
 * Plan: a dQ kernel and a dK/dV kernel that each recompute S and dP (7 GEMMs for the 5 of the algorithm, deterministic, no
 * workspace). The single 5-GEMM kernel of the reference (flash_attn_triton.py:199-226; dQ by load-add-store, here fp32 atomics
 * into a caller-provided accumulator) was built in round 3 and measured slower on MI355X in every form:
 * since round 4 only the developer library carries it. libfasn.so ignores FASN_BWD_ONE_PASS, fasn_bwd_workspace_bytes() returns 0
 * and `workspace` may be NULL; the fields stay in the struct so that the ABI version does not change.
 */
#define FASN_BWD_ONE_PASS 1 /* fasn_bwd_args.flags: reserved (one-pass backward: developer library only); ignored by libfasn.so */
typedef struct fasn_bwd_args {
    fasn_fwd_args fwd; /* same views as forward; o and lse are inputs here */
    fasn_view4 dout;   /* [B,H,Sq,Dv] */
    fasn_view4 dq;     /* [B,H,Sq,D]  out */
    fasn_view4 dk;     /* [B,H/kv_group,Sk,D]  out (summed over the query heads of a GQA group inside the kernel) */
    fasn_view4 dv;     /* [B,H/kv_group,Sk,Dv] out */
    float* delta;      /* [B,H,Sq] fp32 scratch */
    void* workspace;
    size_t workspace_bytes;
    fasn_view4 dbias;  /* optional out (ABI 2): gradient of the additive bias, key stride 1; ptr NULL = not wanted. Needs fwd.bias.
                          Dense form: dS as [B,H,Sq,Sk] in `dtype` (the dQ kernels store it; the caller sums over whatever its
                          bias broadcasts). Reduced form (ABI 4): a batch and / or head stride of 0 (with B > 1 / H > 1) asks for the
                          sum over that dimension - dbias is then [1 or B, 1 or H, Sq, Sk], written once by a kernel that walks
                          the (b,h) sharing each bias tile (csrc/fasn_bwd_dbias_ws.h / fasn_bwd_dbias.h; 16-bit q/k/v, no dropout; no [B,H,Sq,Sk]
                          buffer anywhere). */
    int32_t flags;     /* ABI 4: FASN_BWD_* bits, 0 = default */
    int32_t dbias_dtype; /* ABI 4, reduced form only: FASN_BIAS_SAME (0 means the same) = `dtype`, FASN_BIAS_F32 = fp32 elements */
} fasn_bwd_args;

int fasn_abi_version(void);
const char* fasn_strerror(int code);

/* 1 if (dtype, D, Dv) has a compiled kernel, else 0. */
int fasn_supported(int32_t dtype, int32_t D, int32_t Dv);

int fasn_fwd(const fasn_fwd_args* args, fasn_stream_t stream);

/*
 * Which kernel family `args` is routed to (ABI 4; the backward: fasn_bwd_path below - the same family except at head dim 256): a non-negative
 * FASN_PATH_* value, or a negative FASN_E* code for arguments fasn_fwd would refuse. Every family gives the same results; they
 * differ in speed. FASN_PATH_ELEMENT is the one to know about: masks / biases whose rows cannot be moved in aligned vector
 * pieces (unaligned or strided rows, key stride != 1; an fp32 bias next to 16-bit q at head dim 256, under dropout, or with rows
 * that are not 16-byte aligned - at head dims <= 128 an aligned fp32 bias takes the vector family since ABI 5), scale <= 0 with a bias, fp16 with
 * scale*log2(e) > 8 take per-element loads and run 3-5 x slower than the vector path (ABI 6: dropout at head dim 256 no longer does - it runs the vector general kernels, and reports FASN_PATH_VECTOR, whenever the call's operands allow). Nothing is launched. (The reference has no counterpart: its SDPA backends are picked inside torch.)
 */
#define FASN_PATH_PLAIN 0       /* no mask / bias (causal or not) */
#define FASN_PATH_KEYPAD 1      /* key-padding mask as per-tile visibility bits */
#define FASN_PATH_VECTOR 2      /* mask and / or bias through aligned vector loads / LDS images */
#define FASN_PATH_BIAS_KEYPAD 3 /* vector bias + key-padding visibility bits */
#define FASN_PATH_ELEMENT 4     /* per-element loads (slow) */
#define FASN_PATH_FP32 5        /* fp32 q/k/v: exact-fp32 MFMA kernels (1/16 of the 16-bit rate) */
int fasn_fwd_path(const fasn_fwd_args* args);

/*
 * Forward with a caller-provided workspace: short-query / long-key ("decode") shapes have too few (batch, head, query block)
 * units to fill the GPU, so the keys of each unit are split over several workgroups whose partial results
 * (fp32 accumulator + running max / sum per row) are merged by a second kernel. fasn_fwd_workspace_bytes() returns the
 * bytes that plan needs for `args` (0 = the plain path is used; fasn_fwd_ws then equals fasn_fwd). The workspace must be
 * 16-byte aligned device memory; a NULL or too small workspace silently selects the plain path. Same results either way.
 * Round 5, second use: LONG plain / causal launches at head dim 64 (8+ rounds of workgroups) ask for 64 bytes - eight item counters the
 * library zeroes itself - and then deal their (head, query block) items dynamically across the XCDs of the part (they differ in speed by up
 * to 4 %; a static deal ends with the slowest). Bit-identical results; without the workspace the static deal runs.
 */
size_t fasn_fwd_workspace_bytes(const fasn_fwd_args* args);
int fasn_fwd_ws(const fasn_fwd_args* args, void* workspace, size_t workspace_bytes, fasn_stream_t stream);

/*
 * Dropout stream position in device memory (replaces the host-side philox seed / offset bookkeeping behind
 * flash_attention_softmax_n/core/flash_attn.py:122 and functional.py:92): copies state[0..1] = {seed, offset} to out[0..1]
 * (out may be NULL) and advances state[1] by `increment`, as one tiny kernel on `stream` - capturable, no host round trip.
 * A forward call takes `out` as its rng_state; the backward of that call gets the same `out`.
 */
int fasn_rng_advance(uint64_t* state, uint64_t* out, uint64_t increment, fasn_stream_t stream);

size_t fasn_bwd_workspace_bytes(const fasn_bwd_args* args);
int fasn_bwd(const fasn_bwd_args* args, fasn_stream_t stream);

/*
 * The launches behind a call (ABI 5, the cfg field ABI 6; diagnostic like fasn_fwd_path, no counterpart in the reference, whose launch sites are
 * flash_attention_softmax_n/core/flash_attn_triton.py:278-291,316-335): runs the host side of fasn_fwd (FASN_PLAN_FWD, reads only
 * args->fwd), fasn_bwd (FASN_PLAN_BWD) or fasn_fwd_ws with the workspace it asks for (FASN_PLAN_FWD_WS) on `args` with every launch
 * site recording instead of launching, and writes one line per kernel into `buf`:
 *     "kernel_name<template arguments> grid=G block=T lds=L cfg=C\n"
 * (NUL-terminated; ABI 6: C names the template arguments of the kernel family - "bf16,D=64,QB=2,plain,OCC=2,NW=4,RING=2,SEED=2" for
 * fasn_fwd_kernel<fasn::bf16_tag, 64, 2, 0, 2, 4, 0, 0, 2, 0, 2, 1, 0, 0>: element type, head dim, 32-row blocks per wave, mode, waves per
 * SIMD the kernel is compiled for, waves per workgroup, K/V staging scheme, accumulator seeding; flags that are off are left out; "-" for a
 * kernel without a table - so that a plan, a profile line or a spill table reads without the kernel headers open). No kernel runs, no device memory is touched, no HIP call is made - pointers in `args` only have to be
 * non-NULL and aligned as for the real call. Returns the number of bytes written (without the NUL), the FASN_E* code the real
 * call would return, or FASN_EINVAL when `cap` is too small. The names are those of the code objects inside the library, so a
 * profile (rocprofv3 --kernel-trace) and a register / spill table (llvm-readelf on the bundle) can be matched against them.
 */
#define FASN_PLAN_FWD 0
#define FASN_PLAN_BWD 1
#define FASN_PLAN_FWD_WS 2
int fasn_launch_plan(const fasn_bwd_args* args, int32_t which, char* buf, size_t cap);

/*
 * Which kernel family the BACKWARD of a call takes (ABI 6): the value fasn_fwd_path gives for args->fwd, except that FASN_PATH_ELEMENT is
 * returned whenever the recorded backward plan contains an element-load kernel - today the two only differ through operands the backward's kernels cannot move as vectors although the forward's can (none known: since round 6 the head dim 256 backward
 * has vector instantiations for dense masks and 16-bit biases too); kept so that the front end's slow-path warning follows the backward's launch table, not an assumption about it. Same argument rules as fasn_launch_plan; nothing is launched.
 */
int fasn_bwd_path(const fasn_bwd_args* args);

/*
 * Stand-alone softmax_n over the last dimension of a [rows, cols] matrix (row stride in elements,
 * col stride 1). Replaces flash_attention_softmax_n/core/functional.py:15-29 for device tensors.
 * dtype: FASN_DTYPE_F16 / FASN_DTYPE_BF16 / 2 (= fp32).
 */
#define FASN_DTYPE_F32 2
int fasn_softmax_n_fwd(const void* x, void* y, int64_t rows, int64_t cols, int64_t x_row_stride,
                       int64_t y_row_stride, float n, int32_t dtype, fasn_stream_t stream);
/* dx = y * (dy - sum_j dy_j y_j)  (same formula as softmax; n only enters through y) */
int fasn_softmax_n_bwd(const void* y, const void* dy, void* dx, int64_t rows, int64_t cols,
                       int64_t y_row_stride, int64_t dy_row_stride, int64_t dx_row_stride,
                       int32_t dtype, fasn_stream_t stream);

/*
 * Raw power sums of every row of a [rows, cols] matrix in ONE pass (col stride 1, row stride in elements):
 * sums[row][0..3] += sum x, sum x^2, sum x^3, sum x^4 (fp64; the caller zeroes `sums`). Replaces the repeated
 * mean / subtract / pow passes of flash_attention_softmax_n/analysis/statistics.py:9-79 (variance, skewness, kurtosis of
 * activations) for device tensors. dtype: FASN_DTYPE_F16 / BF16 / F32; rows <= 65535.
 */
int fasn_moments(const void* x, double* sums, int64_t rows, int64_t cols, int64_t row_stride, int32_t dtype,
                 fasn_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* FASN_H_ */

/* attn_n_ref.c — plain-C restatement of attention with softmax_n, fp64 accumulation. TEST INFRASTRUCTURE ONLY.
 *
 * Follows flash_attention_softmax_n/core/functional.py:15-29 (softmax_n: subtract the row max, n*exp(-max) joins the
 * denominator) and :32-93 (scores = q.k*scale + additive term; causal = tril(diagonal=S-L); out = weights @ v),
 * with boolean mask / additive bias given through element strides as flash_attn.py:87-113 combines them.
 * Independent of torch: used by tests as a second oracle ("true" fp64 answer) and by bench.py's cpu_baseline leg.
 * Build: make -C oracle  ->  oracle/libattn_n_ref.so
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>

#ifdef _OPENMP
#include <omp.h>
#endif

/* q [B,H,L,E], k [B,H,S,E], v [B,H,S,Ev] contiguous fp32; out [B,H,L,Ev] fp32; lse [B,H,L] fp32 (may be NULL).
 * mask: uint8 (nonzero = visible) with element strides ms[4] or NULL; bias: fp32 with element strides bs[4] or NULL. */
int attn_n_ref_f32(const float* q, const float* k, const float* v, float* out, float* lse, int B, int H, int L, int S, int E, int Ev,
                   double scale, double n, int causal, const uint8_t* mask, const int64_t* ms, const float* bias, const int64_t* bs) {
    if (!q || !k || !v || !out || B <= 0 || H <= 0 || L <= 0 || S <= 0 || E <= 0 || Ev <= 0) return -1;
    const int off = S - L;
    const int64_t rows = (int64_t)B * H * L;
#pragma omp parallel
    {
        double* x = (double*)malloc(sizeof(double) * (size_t)S);
        double* acc = (double*)malloc(sizeof(double) * (size_t)Ev);
#pragma omp for schedule(dynamic, 16)
        for (int64_t r = 0; r < rows; ++r) {
            const int i = (int)(r % L);
            const int h = (int)((r / L) % H);
            const int b = (int)(r / ((int64_t)L * H));
            const float* qi = q + r * E;
            const float* kb = k + ((int64_t)b * H + h) * S * E;
            const float* vb = v + ((int64_t)b * H + h) * S * Ev;
            double mx = -INFINITY;
            for (int j = 0; j < S; ++j) {
                int show = !(causal && j > i + off);
                if (show && mask) show = mask[b * ms[0] + h * ms[1] + (int64_t)i * ms[2] + (int64_t)j * ms[3]] != 0;
                if (!show) {
                    x[j] = -INFINITY;
                    continue;
                }
                double s = 0.0;
                const float* kj = kb + (int64_t)j * E;
                for (int d = 0; d < E; ++d) s += (double)qi[d] * (double)kj[d];
                s *= scale;
                if (bias) s += (double)bias[b * bs[0] + h * bs[1] + (int64_t)i * bs[2] + (int64_t)j * bs[3]];
                x[j] = s;
                if (s > mx) mx = s;
            }
            if (n > 0.0 && mx < 0.0) mx = 0.0; /* keeps n*exp(-mx) finite; algebraically neutral */
            for (int d = 0; d < Ev; ++d) acc[d] = 0.0;
            double den = 0.0;
            if (mx > -INFINITY) {
                for (int j = 0; j < S; ++j) {
                    if (x[j] == -INFINITY) continue;
                    const double p = exp(x[j] - mx);
                    den += p;
                    const float* vj = vb + (int64_t)j * Ev;
                    for (int d = 0; d < Ev; ++d) acc[d] += p * (double)vj[d];
                }
                den += n * exp(-mx);
            }
            float* o = out + r * Ev;
            for (int d = 0; d < Ev; ++d) o[d] = den > 0.0 ? (float)(acc[d] / den) : 0.0f;
            if (lse) lse[r] = den > 0.0 ? (float)(mx + log(den)) : -INFINITY;
        }
        free(x);
        free(acc);
    }
    return 0;
}

/* softmax_n over the last dimension of a [rows, cols] fp32 matrix (functional.py:15-29) */
int softmax_n_ref_f32(const float* x, float* y, int64_t rows, int64_t cols, double n) {
    if (!x || !y || rows <= 0 || cols <= 0) return -1;
#pragma omp parallel for schedule(static)
    for (int64_t r = 0; r < rows; ++r) {
        const float* xr = x + r * cols;
        float* yr = y + r * cols;
        double mx = -INFINITY;
        for (int64_t c = 0; c < cols; ++c)
            if (xr[c] > mx) mx = xr[c];
        double den = 0.0;
        for (int64_t c = 0; c < cols; ++c) den += exp((double)xr[c] - mx);
        den += n * exp(-mx);
        for (int64_t c = 0; c < cols; ++c) yr[c] = (float)(exp((double)xr[c] - mx) / den);
    }
    return 0;
}

int attn_n_ref_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

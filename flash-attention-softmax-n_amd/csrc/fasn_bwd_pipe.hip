// Software-pipelined backward kernels (fasn_bwd_pipe.h), D = 64 plain / causal: launch plumbing.
#include "fasn_bwd_launch.h"
#include "fasn_bwd_pipe.h"
namespace fasn {

template <typename Tag, int MODE, int DROP>
static int launch_dkdv_pipe(BwdParams p, hipStream_t s) {
    constexpr int BN = 128;
    constexpr int smem = pipe_dkdv_smem_bytes();
    const int nbh = p.f.B * p.f.H;
    p.nblk = (p.f.Sk + BN - 1) / BN;
    constexpr auto kern = &fasn_bwd_dkdv_pipe_kernel<Tag, MODE, DROP>;
    ensure_smem<kern>(smem);
    p.f.pair = (MODE == MODE_CAUSAL && p.nblk > 1 && pair_wanted((long)p.nblk * nbh, wg_slots(2, 4, smem), true)) ? 1 : 0;
    FASN_LAUNCH(kern, dim3((unsigned)((p.f.pair ? (p.nblk + 1) / 2 : p.nblk) * nbh)), dim3(256), smem, s, p);
    return launch_rc();
}

template <typename Tag, int MODE, int DROP>
static int launch_dq_pipe(BwdParams p, hipStream_t s) {
    constexpr int BM = 128;
    constexpr int smem = pipe_dq_smem_bytes();
    const int nbh = p.f.B * p.f.H;
    p.nblk = (p.f.Sq + BM - 1) / BM;
    constexpr auto kern = &fasn_bwd_dq_pipe_kernel<Tag, MODE, DROP>;
    ensure_smem<kern>(smem);
    p.f.pair = (MODE == MODE_CAUSAL && p.nblk > 1 && pair_wanted((long)p.nblk * nbh, wg_slots(2, 4, smem), true)) ? 1 : 0;
    FASN_LAUNCH(kern, dim3((unsigned)((p.f.pair ? (p.nblk + 1) / 2 : p.nblk) * nbh)), dim3(256), smem, s, p);
    return launch_rc();
}

template <typename Tag, int DROP>
static int dq_pipe_mode(const BwdParams& p, int mode, hipStream_t s) {
    return mode == MODE_CAUSAL ? launch_dq_pipe<Tag, MODE_CAUSAL, DROP>(p, s) : launch_dq_pipe<Tag, MODE_PLAIN, DROP>(p, s);
}
int launch_bwd_dq_pipe_d64(const BwdParams& p, const FwdLaunch& l, hipStream_t s) {
    if (p.f.drop_thr) return l.dtype == 1 ? dq_pipe_mode<bf16_tag, 1>(p, l.mode, s) : dq_pipe_mode<f16_tag, 1>(p, l.mode, s);
    return l.dtype == 1 ? dq_pipe_mode<bf16_tag, 0>(p, l.mode, s) : dq_pipe_mode<f16_tag, 0>(p, l.mode, s);
}

#ifdef FASN_DEV_VARIANTS   // one wave per SIMD, 64 keys per wave: measured slower (1241 against 1031 us at M0), developer library only
template <typename Tag, int MODE, int KB>
static int launch_dkdv_pipe2(BwdParams p, hipStream_t s) {
    constexpr int BN = 4 * KB * 32;
    constexpr int smem = pipe_dkdv_smem_bytes();
    const int nbh = p.f.B * p.f.H;
    p.nblk = (p.f.Sk + BN - 1) / BN;
    constexpr auto kern = &fasn_bwd_dkdv_pipe2_kernel<Tag, MODE, KB>;
    ensure_smem<kern>(smem);
    p.f.pair = (MODE == MODE_CAUSAL && p.nblk > 1 && pair_wanted((long)p.nblk * nbh, wg_slots(1, 4, smem), true)) ? 1 : 0;
    FASN_LAUNCH(kern, dim3((unsigned)((p.f.pair ? (p.nblk + 1) / 2 : p.nblk) * nbh)), dim3(256), smem, s, p);
    return launch_rc();
}
int launch_bwd_dkdv_pipe2_d64(const BwdParams& p, const FwdLaunch& l, hipStream_t s) {
    if (l.mode == MODE_CAUSAL) return l.dtype == 1 ? launch_dkdv_pipe2<bf16_tag, MODE_CAUSAL, 2>(p, s) : launch_dkdv_pipe2<f16_tag, MODE_CAUSAL, 2>(p, s);
    return l.dtype == 1 ? launch_dkdv_pipe2<bf16_tag, MODE_PLAIN, 2>(p, s) : launch_dkdv_pipe2<f16_tag, MODE_PLAIN, 2>(p, s);
}
#endif

template <typename Tag, int DROP>
static int dkdv_pipe_mode(const BwdParams& p, int mode, hipStream_t s) {
    return mode == MODE_CAUSAL ? launch_dkdv_pipe<Tag, MODE_CAUSAL, DROP>(p, s) : launch_dkdv_pipe<Tag, MODE_PLAIN, DROP>(p, s);
}
int launch_bwd_dkdv_pipe_d64(const BwdParams& p, const FwdLaunch& l, hipStream_t s) {
    if (p.f.drop_thr) return l.dtype == 1 ? dkdv_pipe_mode<bf16_tag, 1>(p, l.mode, s) : dkdv_pipe_mode<f16_tag, 1>(p, l.mode, s);
    return l.dtype == 1 ? dkdv_pipe_mode<bf16_tag, 0>(p, l.mode, s) : dkdv_pipe_mode<f16_tag, 0>(p, l.mode, s);
}
}  // namespace fasn

// fp32 attention-softmax_n kernels (exact fp32 MFMA): instantiations and launchers.
#include "fasn_f32_kernels.h"
#include "fasn_launch.h"
#include "fasn_bwd_launch.h"

namespace fasn {

template <int D, int MODE>
static int fwd_one(FwdParams p, hipStream_t s) {
    constexpr int smem = 4 * 64 * D * 4;
    p.nqblk = (p.Sq + 127) / 128;
    constexpr auto kern = &fasn_f32_fwd_kernel<D, MODE>;
    ensure_smem<kern>(smem);
    FASN_LAUNCH(kern, dim3((unsigned)(p.nqblk * p.B * p.H)), dim3(256), smem, s, p);
    return launch_rc();
}

template <int D, int MODE>
static int bwd_one(BwdParams p, hipStream_t s) {
    const int nbh = p.f.B * p.f.H;
    {
        constexpr int RPB = 256 / (D / 4);
        const int64_t rows = (int64_t)nbh * p.f.Sq;
        FASN_LAUNCH((fasn_f32_delta_kernel<D>), dim3((unsigned)((rows + RPB - 1) / RPB)), dim3(256), 0, s, p);
    }
    {
        constexpr int smem = 4 * 64 * D * 4;
        p.nblk = (p.f.Sq + 127) / 128;
        constexpr auto kern = &fasn_f32_dq_kernel<D, MODE>;
        ensure_smem<kern>(smem);
        FASN_LAUNCH(kern, dim3((unsigned)(p.nblk * nbh)), dim3(256), smem, s, p);
    }
    {
        constexpr int smem = 4 * 64 * D * 4 + 4 * 64 * 4;
        p.nblk = (p.f.Sk + 127) / 128;
        constexpr auto kern = &fasn_f32_dkdv_kernel<D, MODE>;
        ensure_smem<kern>(smem);
        // one workgroup per K/V head: the kernel sums the query heads of a GQA group itself
        FASN_LAUNCH(kern, dim3((unsigned)(p.nblk * (nbh / p.f.kvg))), dim3(256), smem, s, p);
    }
    return launch_rc();
}

int launch_fwd_f32(const FwdParams& p, const FwdLaunch& l, hipStream_t s) {
    const bool c = l.mode == MODE_CAUSAL;
    if (l.mode >= MODE_GENERAL || p.drop_thr) {   // mask / bias / dropout: element-load instantiation
        switch (l.D) {
            case 32: return fwd_one<32, MODE_GENERAL_SLOW>(p, s);
            case 64: return fwd_one<64, MODE_GENERAL_SLOW>(p, s);
            case 128: return fwd_one<128, MODE_GENERAL_SLOW>(p, s);
            default: return -3;
        }
    }
    switch (l.D) {
        case 32: return c ? fwd_one<32, MODE_CAUSAL>(p, s) : fwd_one<32, MODE_PLAIN>(p, s);
        case 64: return c ? fwd_one<64, MODE_CAUSAL>(p, s) : fwd_one<64, MODE_PLAIN>(p, s);
        case 128: return c ? fwd_one<128, MODE_CAUSAL>(p, s) : fwd_one<128, MODE_PLAIN>(p, s);
        default: return -3;
    }
}

int launch_bwd_f32(const BwdParams& p, const FwdLaunch& l, hipStream_t s) {
    const bool c = l.mode == MODE_CAUSAL;
    if (l.mode >= MODE_GENERAL || p.f.drop_thr) {
        switch (l.D) {
            case 32: return bwd_one<32, MODE_GENERAL_SLOW>(p, s);
            case 64: return bwd_one<64, MODE_GENERAL_SLOW>(p, s);
            case 128: return bwd_one<128, MODE_GENERAL_SLOW>(p, s);
            default: return -3;
        }
    }
    switch (l.D) {
        case 32: return c ? bwd_one<32, MODE_CAUSAL>(p, s) : bwd_one<32, MODE_PLAIN>(p, s);
        case 64: return c ? bwd_one<64, MODE_CAUSAL>(p, s) : bwd_one<64, MODE_PLAIN>(p, s);
        case 128: return c ? bwd_one<128, MODE_CAUSAL>(p, s) : bwd_one<128, MODE_PLAIN>(p, s);
        default: return -3;
    }
}

}  // namespace fasn

// D = 64 backward instantiations: <QB (dQ: 32-row blocks/wave), KB (dK/dV: 32-key blocks/wave), occupancies>
// (the two-wave kernels of D = 128 measured slower here: (8,16,4096,64) backward 2.18 ms against 1.81 ms - at D = 64 the one-wave
// kernels already run two waves per SIMD and the exponentials, not registers, are the limit)
#include "fasn_bwd_launch.h"
#ifndef FASN_DELTA_KERNEL
#define FASN_DELTA_KERNEL 0   // (round 5 A/B: 1 = keep the separate delta launch in front of the pipelined kernels)
#endif
namespace fasn {
#ifdef FASN_DEV_VARIANTS
int launch_bwd_d64_exp(const BwdParams& p, int which, hipStream_t s);
#endif
int launch_bwd_d64(const BwdParams& p, const FwdLaunch& l, hipStream_t s) {
#ifdef FASN_DEV_VARIANTS
    if (((FASN_BWD_VARIANT >> 8) & 3) && l.mode == MODE_PLAIN && l.dtype == 1) return launch_bwd_d64_exp(p, (FASN_BWD_VARIANT >> 8) & 3, s);
#endif
#ifdef FASN_DEV_VARIANTS
    if (p.dqacc != nullptr)   // developer library: fasn_api.hip sets the accumulator only where the one-pass backward applies
        return launch_bwd_fused_d64(p, l, s);
#endif
    // plain / causal without grouped K/V: the software-pipelined kernels of fasn_bwd_pipe.h
    // (developer library: bwd_variant bit 6 / bit 7 = the round-3 dK/dV / dQ kernel instead, for same-box A/B)
    BwdParams q = p;
    q.skip = 0;
    const bool pipe_ok = (l.mode == MODE_PLAIN || l.mode == MODE_CAUSAL) && p.f.kvg == 1;   // (with or without dropout)
    if (pipe_ok && !(FASN_BWD_VARIANT & 64)) q.skip |= 1;
    if (pipe_ok && !(FASN_BWD_VARIANT & 128)) q.skip |= 2;
    // both pipelined kernels: no delta launch either - the dQ kernel, which runs first, computes delta = rowsum(O o dO) of its rows in its prologue
    // (bit-identical to fasn_bwd_delta_kernel) and stores it for the dK/dV kernel (C2 backward -7 %, the other D = 64 configs -1.5 .. -2.5 %)
    if (q.skip == 3 && !FASN_DELTA_KERNEL) q.skip |= 4;
    int rc = l.dtype == 1 ? launch_bwd_mode<bf16_tag, 64, 1, 1, 2, 2>(q, l.mode, s) : launch_bwd_mode<f16_tag, 64, 1, 1, 2, 2>(q, l.mode, s);
    if (rc) return rc;
    if (q.skip & 2) rc = launch_bwd_dq_pipe_d64(p, l, s);
    if (rc) return rc;
#ifdef FASN_DEV_VARIANTS
    if ((q.skip & 1) && (FASN_BWD_VARIANT & 1024)) return launch_bwd_dkdv_pipe2_d64(p, l, s);   // one wave per SIMD, 64 keys per wave
#endif
    if (q.skip & 1) rc = launch_bwd_dkdv_pipe_d64(p, l, s);
    return rc;
}
}  // namespace fasn

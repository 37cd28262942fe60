"""Argument blocks of the BASELINE.json configs without a GPU: shapes, strides and modes as bench.py / the front end fill them, dummy
(aligned, non-NULL) addresses - enough for fasn_launch_plan, which launches nothing. Used by the CPU tests that ask the library which
kernels a config reaches (tests/test_abi_cpu.py, tests/test_spill_gate.py)."""

DUMMY = 1 << 20   # any 16-byte aligned non-NULL address: the launch recorder touches no memory

# name: (B, H, S, D, dtype enum (0 f16, 1 bf16, 2 f32), n, causal, bias [H,L,S] + key-padding mask [B,1,1,S])
CONFIGS = {
    "c1": (2, 2, 128, 32, 2, 1.0, 0, False),
    "m0": (8, 16, 4096, 64, 1, 1.0, 0, False),
    "c2": (8, 16, 1024, 64, 1, 1.0, 0, False),
    "c3": (8, 16, 4096, 64, 0, 1.0, 1, False),
    "c4": (4, 32, 8192, 128, 1, 0.5, 0, True),
    "c5": (64, 16, 4096, 64, 1, 1.0, 1, False),
}


def _view(v, strides, ptr=DUMMY):
    v.ptr = ptr
    for i, s in enumerate(strides):
        v.stride[i] = s


def bwd_args(pkg, name):
    """BwdArgs of BASELINE config `name` (contiguous [B,H,S,D] tensors); .fwd is what fasn_fwd gets. `name` may also be a tuple of the CONFIGS form."""
    B, H, S, D, dt, n, causal, c4 = CONFIGS[name] if isinstance(name, str) else name
    a = pkg._lib.BwdArgs()
    f = a.fwd
    dense = (H * S * D, S * D, D, 1)
    for v in (f.q, f.k, f.v, f.o, a.dout, a.dq, a.dk, a.dv):
        _view(v, dense)
    f.lse = DUMMY
    a.delta = DUMMY
    f.dtype, f.B, f.H, f.Sq, f.Sk, f.D, f.Dv = dt, B, H, S, S, D, D
    f.scale, f.softmax_n, f.causal = 1.0 / D ** 0.5, n, causal
    if c4:
        _view(f.mask, (S, 0, 0, 1))             # [B,1,1,S] boolean key-padding mask, expanded by stride 0
        _view(f.bias, (0, S * S, S, 1))         # [H,L,S] bias, broadcast over the batch
        f.bias_dtype = pkg._lib.FASN_BIAS_SAME
    return a


def kernels(pkg, name, which):
    """[(kernel<template arguments>, grid, block, lds)] of config `name`, which = 'fwd' | 'bwd'"""
    code = {"fwd": pkg._lib.FASN_PLAN_FWD_WS, "bwd": pkg._lib.FASN_PLAN_BWD}[which]
    return pkg._lib.launch_plan(bwd_args(pkg, name), code)

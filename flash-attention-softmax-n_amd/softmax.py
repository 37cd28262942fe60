"""softmax_n for device tensors on the HIP row kernel (fasn_softmax.hip).

Mirrors flash_attention_softmax_n/core/functional.py:15-29: softmax_n(x, n=None, dim=None, dtype=None).
"""
from typing import Optional

import torch
from torch import Tensor

from . import _lib

_DT = {torch.float16: _lib.FASN_DTYPE_F16, torch.bfloat16: _lib.FASN_DTYPE_BF16, torch.float32: _lib.FASN_DTYPE_F32}


class _SoftmaxN(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x2d: Tensor, n: float):
        lib = _lib.load()
        y = torch.empty_like(x2d)
        rows, cols = x2d.shape
        with torch.cuda.device(x2d.device):
            rc = lib.fasn_softmax_n_fwd(x2d.data_ptr(), y.data_ptr(), rows, cols, x2d.stride(0), y.stride(0), n, _DT[x2d.dtype],
                                        torch.cuda.current_stream(x2d.device).cuda_stream)
        _lib.check(rc, "fasn_softmax_n_fwd")
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, dy: Tensor):
        lib = _lib.load()
        (y,) = ctx.saved_tensors
        dy = dy.contiguous()
        dx = torch.empty_like(y)
        rows, cols = y.shape
        with torch.cuda.device(y.device):
            rc = lib.fasn_softmax_n_bwd(y.data_ptr(), dy.data_ptr(), dx.data_ptr(), rows, cols, y.stride(0), dy.stride(0), dx.stride(0),
                                        _DT[y.dtype], torch.cuda.current_stream(y.device).cuda_stream)
        _lib.check(rc, "fasn_softmax_n_bwd")
        return dx, None


def softmax_n(x: Tensor, n: Optional[float] = None, dim: Optional[int] = None, dtype=None) -> Tensor:
    """softmax_n(x)_i = exp(x_i) / (n + sum_j exp(x_j)) along `dim` (default -1); output cast to `dtype` if given."""
    if not x.is_cuda:
        raise RuntimeError("softmax_n: device tensors only (no CPU fallback)")
    if x.dtype not in _DT:
        raise NotImplementedError(f"softmax_n: dtype {x.dtype} not supported")
    n = 0.0 if n is None else float(n)
    dim = -1 if dim is None else dim
    xt = x.movedim(dim, -1)
    shape = xt.shape
    x2d = xt.reshape(-1, shape[-1])
    if x2d.stride(-1) != 1:
        x2d = x2d.contiguous()
    y = _SoftmaxN.apply(x2d, n).reshape(shape).movedim(-1, dim)
    return y if dtype is None else y.type(dtype)

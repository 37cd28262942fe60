"""Moments of device tensors in one pass over the data (HIP kernel fasn_moments).

Same functions as the reference's flash_attention_softmax_n/analysis/statistics.py:9-79 (used there to measure activation
and weight outliers: variance, skewness, excess kurtosis, per sample or over given dims). The reference runs mean / subtract /
pow / mean for every statistic; here one kernel reads each element once and returns the raw power sums in fp64, and the
central moments follow from them. Only orders k <= 4 exist (what the reference's own statistics use).
"""
from typing import Optional, Tuple, Union

import torch
from torch import Tensor

from . import _lib

_Dim = Optional[Union[int, Tuple[int, ...]]]
_DT = {torch.float16: _lib.FASN_DTYPE_F16, torch.bfloat16: _lib.FASN_DTYPE_BF16, torch.float32: _lib.FASN_DTYPE_F32}
_MAX_ROWS = 65535


def _power_sums(x: Tensor, dim: _Dim):
    """returns (count per output element, fp64 sums [rows, 4], output shape)"""
    if not x.is_cuda:
        raise RuntimeError("statistics: device tensors only (no CPU fallback)")
    if x.dtype not in _DT:
        raise NotImplementedError(f"statistics: dtype {x.dtype} not supported")
    nd = x.dim()
    dims = tuple(range(nd)) if dim is None else ((dim,) if isinstance(dim, int) else tuple(dim))
    dims = tuple(sorted(d % nd for d in dims))
    keep = tuple(d for d in range(nd) if d not in dims)
    out_shape = tuple(x.shape[d] for d in keep)
    xt = x.permute(*keep, *dims)
    rows = 1
    for d in keep:
        rows *= x.shape[d]
    cols = x.numel() // max(rows, 1)
    x2d = xt.reshape(rows, cols)
    if x2d.stride(-1) != 1 or (rows > 1 and x2d.stride(0) < cols):
        x2d = x2d.contiguous()
    sums = torch.zeros((rows, 4), dtype=torch.float64, device=x.device)
    lib = _lib.load()
    stream = torch.cuda.current_stream(x.device).cuda_stream
    with torch.cuda.device(x.device):
        for r0 in range(0, rows, _MAX_ROWS):
            r1 = min(rows, r0 + _MAX_ROWS)
            part = x2d[r0:r1]
            _lib.check(lib.fasn_moments(part.data_ptr(), sums[r0:r1].data_ptr(), r1 - r0, cols,
                                        part.stride(0) if r1 - r0 > 1 else cols, _DT[x.dtype], stream), "fasn_moments")
    return cols, sums, out_shape


def _central(sums: Tensor, count: int, k: int) -> Tensor:
    s1, s2, s3, s4 = (sums[:, i] / count for i in range(4))
    if k == 1:
        return torch.zeros_like(s1)
    if k == 2:
        return s2 - s1 * s1
    if k == 3:
        return s3 - 3 * s1 * s2 + 2 * s1 ** 3
    if k == 4:
        return s4 - 4 * s1 * s3 + 6 * s1 * s1 * s2 - 3 * s1 ** 4
    raise NotImplementedError("central moments of order > 4 are not provided (one pass keeps four power sums)")


@torch.no_grad()
def central_moment(x: Tensor, k: int, dim: _Dim = None) -> Tensor:
    """k-th moment about the mean (reference statistics.py:9-14), k in 1..4."""
    count, sums, shape = _power_sums(x, dim)
    return _central(sums, count, int(k)).reshape(shape).to(x.dtype)


@torch.no_grad()
def variance(x: Tensor, dim: _Dim = None) -> Tensor:
    return central_moment(x, 2, dim=dim)


@torch.no_grad()
def standard_deviation(x: Tensor, dim: _Dim = None) -> Tensor:
    count, sums, shape = _power_sums(x, dim)
    return _central(sums, count, 2).clamp_min(0).sqrt().reshape(shape).to(x.dtype)


@torch.no_grad()
def standardized_moment(x: Tensor, k: int, dim: _Dim = None) -> Tensor:
    """central moment k divided by variance^(k/2) (reference statistics.py:27-32)"""
    count, sums, shape = _power_sums(x, dim)
    return (_central(sums, count, int(k)) / _central(sums, count, 2) ** (k / 2)).reshape(shape).to(x.dtype)


@torch.no_grad()
def skewness(x: Tensor, dim: _Dim = None) -> Tensor:
    return standardized_moment(x, 3, dim=dim)


@torch.no_grad()
def kurtosis(x: Tensor, dim: _Dim = None) -> Tensor:
    """excess kurtosis (reference statistics.py:40-45)"""
    count, sums, shape = _power_sums(x, dim)
    return (_central(sums, count, 4) / _central(sums, count, 2) ** 2 - 3.0).reshape(shape).to(x.dtype)


def _sample_dims(x: Tensor) -> Tuple[int, ...]:
    return tuple(range(1, x.ndim))


@torch.no_grad()
def variance_batch_mean(x: Tensor) -> float:
    """variance of every sample of the batch, averaged over the batch (reference statistics.py:55-61)"""
    count, sums, _ = _power_sums(x, _sample_dims(x))
    return _central(sums, count, 2).mean().item()


@torch.no_grad()
def skewness_batch_mean(x: Tensor) -> float:
    count, sums, _ = _power_sums(x, _sample_dims(x))
    return (_central(sums, count, 3) / _central(sums, count, 2) ** 1.5).mean().item()


@torch.no_grad()
def kurtosis_batch_mean(x: Tensor) -> float:
    count, sums, _ = _power_sums(x, _sample_dims(x))
    return (_central(sums, count, 4) / _central(sums, count, 2) ** 2 - 3.0).mean().item()

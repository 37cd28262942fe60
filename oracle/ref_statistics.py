"""TEST INFRASTRUCTURE ONLY (see oracle/ref_attention.py): fp64 numpy restatement of the reference's moment statistics,
flash_attention_softmax_n/analysis/statistics.py:9-79 — central_moment(x,k,dim) = mean((x - mean(x))^k), standardized moment =
central_k / variance^(k/2), skewness (k=3), excess kurtosis (k=4, minus 3), and the per-sample "batch mean" variants (:48-79)."""
import numpy as np


def central_moment(x, k, dim=None):
    x = np.asarray(x, dtype=np.float64)
    return np.mean((x - np.mean(x, axis=dim, keepdims=True)) ** k, axis=dim)


def variance(x, dim=None):
    return central_moment(x, 2, dim)


def standardized_moment(x, k, dim=None):
    return central_moment(x, k, dim) / variance(x, dim) ** (k / 2)


def skewness(x, dim=None):
    return standardized_moment(x, 3, dim)


def kurtosis(x, dim=None):
    return standardized_moment(x, 4, dim) - 3.0


def batch_mean(stat, x):
    x = np.asarray(x)
    return float(np.mean(stat(x, tuple(range(1, x.ndim)))))

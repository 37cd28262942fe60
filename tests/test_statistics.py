"""Moment statistics (reference analysis/statistics.py): the numpy oracle against the golden outputs of the real reference
(CPU), and the one-pass HIP kernel against both (-m gpu)."""
import os

import numpy as np
import pytest
import torch

from oracle import ref_statistics as rs

import flash_attention_softmax_n_amd.synth as synth

DIMS = {"all": None, "last": -1, "first": 0}


def _inputs(g, name):
    shape = tuple(int(v) for v in g[f"{name}_shape"])
    x = synth.counter_normal(shape, int(g[f"{name}_seed"]), std=1.5, dtype=torch.float32) + 0.3
    if int(g[f"{name}_outliers"]):
        x.view(-1)[::97] *= 25.0
    assert synth.checksum(x) == int(g[f"{name}_checksum"])
    return x


@pytest.mark.parametrize("name", ["a", "b"])
def test_numpy_oracle_matches_reference_outputs(golden_dir, name):
    g = np.load(os.path.join(golden_dir, "g6_statistics.npz"))
    x = _inputs(g, name).double().numpy()
    dims = dict(DIMS, sample=tuple(range(1, x.ndim)))
    for dn, dim in dims.items():
        np.testing.assert_allclose(rs.variance(x, dim), g[f"{name}_var_{dn}"], rtol=1e-10)
        np.testing.assert_allclose(rs.skewness(x, dim), g[f"{name}_skew_{dn}"], rtol=1e-9, atol=1e-12)
        np.testing.assert_allclose(rs.kurtosis(x, dim), g[f"{name}_kurt_{dn}"], rtol=1e-9, atol=1e-12)
        np.testing.assert_allclose(rs.central_moment(x, 3, dim), g[f"{name}_m3_{dn}"], rtol=1e-9, atol=1e-12)
    assert abs(rs.batch_mean(rs.variance, x) - float(g[f"{name}_var_bm"])) < 1e-10
    assert abs(rs.batch_mean(rs.skewness, x) - float(g[f"{name}_skew_bm"])) < 1e-10
    assert abs(rs.batch_mean(rs.kurtosis, x) - float(g[f"{name}_kurt_bm"])) < 1e-10


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["a", "b"])
def test_one_pass_kernel_matches_reference_outputs(pkg, dev, golden_dir, name):
    st = pkg.statistics
    g = np.load(os.path.join(golden_dir, "g6_statistics.npz"))
    x = _inputs(g, name).to(dev)
    dims = dict(DIMS, sample=tuple(range(1, x.ndim)))
    for dn, dim in dims.items():
        for fn, key in ((st.variance, "var"), (st.skewness, "skew"), (st.kurtosis, "kurt"), (lambda t, dim: st.central_moment(t, 3, dim=dim), "m3")):
            got = fn(x, dim=dim).double().cpu().numpy()
            want = g[f"{name}_{key}_{dn}"]
            assert got.shape == want.shape
            np.testing.assert_allclose(got, want, rtol=2e-5, atol=2e-5 * max(1.0, float(np.abs(want).max())))
    assert abs(st.variance_batch_mean(x) - float(g[f"{name}_var_bm"])) < 1e-5 * max(1.0, abs(float(g[f"{name}_var_bm"])))
    assert abs(st.skewness_batch_mean(x) - float(g[f"{name}_skew_bm"])) < 1e-5 * max(1.0, abs(float(g[f"{name}_skew_bm"])))
    assert abs(st.kurtosis_batch_mean(x) - float(g[f"{name}_kurt_bm"])) < 1e-5 * max(1.0, abs(float(g[f"{name}_kurt_bm"])))


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16, torch.float32])
def test_one_pass_kernel_layouts_and_sizes(pkg, dev, dtype):
    """large rows (several chunks per row), many short rows, strided inputs, 16-bit inputs: against the numpy oracle on the same
    (already rounded) values"""
    st = pkg.statistics
    for shape, dim in (((2, 1 << 20), -1), ((5000, 37), -1), ((8, 16, 300, 64), (1, 2, 3)), ((6, 40, 50), 1), ((3, 7, 11, 13), (0, 2))):
        x = (synth.counter_normal(shape, 7, std=2.0, dtype=torch.float32) + 0.5).to(dtype).to(dev)
        xs = x.double().cpu().numpy()
        d = dim if isinstance(dim, tuple) else (dim,)
        for fn, ofn in ((st.variance, rs.variance), (st.skewness, rs.skewness), (st.kurtosis, rs.kurtosis)):
            got = fn(x, dim=dim).double().cpu().numpy()
            want = ofn(xs, d)
            tol = {torch.float32: 2e-5, torch.float16: 2e-3, torch.bfloat16: 1.6e-2}[dtype]   # output is cast to x.dtype
            np.testing.assert_allclose(got, want, rtol=tol, atol=tol)
    with pytest.raises(NotImplementedError):
        st.central_moment(x, 5)
    with pytest.raises(RuntimeError):
        st.variance(x.cpu())

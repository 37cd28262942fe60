"""Pin the oracle (oracle/ref_attention.py torch restatement + oracle/attn_n_ref.c) against outputs of the REAL reference
stored in tests/golden (made by tests/golden/make_golden.py) and against the reference tests' closed forms. CPU only."""
import os

import numpy as np
import pytest
import torch

from oracle import c_oracle
from oracle.ref_attention import (analytic_answer, analytic_causal_answer, ref_attention_n, ref_attention_n_rows, ref_softmax_n)

import flash_attention_softmax_n_amd.synth as synth

NS = (0.0, 0.5, 1.0, 4.0)


def _g(golden_dir, name):
    return np.load(os.path.join(golden_dir, name))


@pytest.mark.parametrize("n", NS)
@pytest.mark.parametrize("causal", [False, True])
def test_g1_forward_backward_fp32(golden_dir, n, causal):
    g = _g(golden_dir, "g1_c1.npz")
    tag = f"n{n}_c{int(causal)}"
    q, k, v = (torch.from_numpy(g[x]).requires_grad_() for x in ("q", "k", "v"))
    o = ref_attention_n(q, k, v, softmax_n_param=n, is_causal=causal)
    o.backward(torch.from_numpy(g["dout"]))
    # same ops in the same order as the reference: agreement to fp32 rounding (reference CPU tests use atol 1e-6)
    assert np.abs(o.detach().numpy() - g[f"o_{tag}"]).max() <= 1e-6
    for name, t in (("dq", q), ("dk", k), ("dv", v)):
        assert np.abs(t.grad.numpy() - g[f"{name}_{tag}"]).max() <= 2e-6
    # independent C oracle (fp64 accumulation)
    oc = c_oracle.attention_n(g["q"], g["k"], g["v"], n=n, causal=causal)
    assert np.abs(oc - g[f"o_{tag}"]).max() <= 2e-6


@pytest.mark.parametrize("n", NS)
@pytest.mark.parametrize("causal", [False, True])
def test_g1_native_bf16_matches_reference_numerics(golden_dir, n, causal):
    g = _g(golden_dir, "g1_c1.npz")
    q, k, v = (torch.from_numpy(g[x]).bfloat16() for x in ("q", "k", "v"))
    o = ref_attention_n(q, k, v, softmax_n_param=n, is_causal=causal).float().numpy()
    # same eager bf16 op sequence as the reference -> identical up to 1 bf16 ulp of O (~2^-8 * 0.2)
    assert np.abs(o - g[f"o_bf16native_n{n}_c{int(causal)}"]).max() <= 2e-3


@pytest.mark.parametrize("n", [0.0, 1.0, 4.0])
@pytest.mark.parametrize("causal", [False, True])
def test_g1_reference_flash_cpu_path_agrees(golden_dir, n, causal):
    """the reference's own flash_attention_n (SDPA on zero-padded K/V, flash_attn.py:66-124) == its slow path == oracle"""
    g = _g(golden_dir, "g1_c1.npz")
    tag = f"n{n}_c{int(causal)}"
    assert np.abs(g[f"o_flashcpu_{tag}"] - g[f"o_{tag}"]).max() <= 1e-6
    oc = c_oracle.attention_n(g["q"], g["k"], g["v"], n=n, causal=causal)
    assert np.abs(oc - g[f"o_flashcpu_{tag}"]).max() <= 2e-6


@pytest.mark.parametrize("n", [0.0, 1.0, 1e-3, 1e-6, 4.0])
def test_g3_softmax_kat(golden_dir, n):
    g = _g(golden_dir, "g3_softmax.npz")
    x = torch.from_numpy(g["x"])
    y = ref_softmax_n(x, n=n).numpy()
    assert np.abs(y - g[f"y_n{n}"]).max() <= 1e-7
    num = g["numerators"]
    want = num / (n + num.sum(-1, keepdims=True))          # test_functional.py:15-31
    assert np.allclose(y, want, rtol=1e-6, atol=0)
    assert np.allclose(c_oracle.softmax_n(g["x"], n), want, rtol=1e-6, atol=0)
    big = ref_softmax_n(torch.from_numpy(g["big"]), n=n)   # exp([12, 89, 710]) overflows naively (test_functional.py:33-36)
    assert big.sum().item() == 1.0
    assert np.array_equal(big.numpy(), g[f"ybig_n{n}"])


@pytest.mark.parametrize("n", [0.0, 1.0, 1e-3, 1e-6, 4.0])
@pytest.mark.parametrize("weight", [10, 1, 0.1, -0.1, -1])
def test_closed_form_small(n, weight):
    """reference tests/cpu/core/test_functional.py:126-149 (N=2, L=3, S=4, E=8, Ev=7, scale=0.3)"""
    N, L, S, E, Ev, scale = 2, 3, 4, 8, 7, 0.3
    q, k, v = weight * torch.ones(N, L, E), weight * torch.ones(N, S, E), weight * torch.ones(N, S, Ev)
    a = ref_attention_n(q, k, v, scale=scale, softmax_n_param=n)
    assert torch.allclose(a, torch.full_like(a, analytic_answer(weight, S, E, scale, n)), rtol=1e-5, atol=1e-6)
    b = ref_attention_n(q, k, v, scale=scale, softmax_n_param=n, is_causal=True)
    want = torch.tensor(analytic_causal_answer(weight, L, S, E, scale, n))
    assert torch.allclose(b[0, :, 0], want, rtol=1e-5, atol=1e-6)
    c = c_oracle.attention_n(q[:, None].numpy(), k[:, None].numpy(), v[:, None].numpy(), n=n, scale=scale, causal=True)
    assert np.allclose(c[0, 0, :, 0], want.numpy(), rtol=1e-5, atol=1e-6)


def _head_slice(name, shape, dtype, b, h):
    B, H, S, D = shape
    seeds = {"q": 101, "k": 102, "v": 103, "dout": 104}
    return synth.counter_normal((S, D), seeds[name], dtype=dtype, start=((b * H + h) * S) * D, std=1.0 if name == "dout" else 0.5)


G4 = {"c2": torch.bfloat16, "c3": torch.float16, "m0": torch.bfloat16, "c4": torch.bfloat16, "c5": torch.bfloat16}


@pytest.mark.parametrize("cfg", ["c2", "c3", "m0", "c4", "c5"])
def test_g4_generator_bits_and_sampled_rows(golden_dir, cfg):
    """the counter-based generator reproduces the fixture's inputs bit for bit, and the oracle reproduces the reference's rows"""
    g = _g(golden_dir, f"g4_{cfg}.npz")
    dtype = G4[cfg]
    B, H, S, D = (int(x) for x in g["shape"])
    n, causal = float(g["n"]), bool(g["causal"])
    rows = torch.from_numpy(g["rows"])
    heads = g["heads"][:2] if S > 4096 else g["heads"]  # keep the CPU suite short
    for hi, (b, h) in enumerate(heads):
        b, h = int(b), int(h)
        q, k, v = (_head_slice(nm, (B, H, S, D), dtype, b, h) for nm in ("q", "k", "v"))
        assert [synth.checksum(q), synth.checksum(k), synth.checksum(v)] == list(g["checksums"][hi])
        bias = mask = None
        if cfg == "c4":
            bias = synth.alibi_bias_rows(H, S, S, [h], g["rows"], dtype)[0]
            mask = synth.keypad_mask(B, S)[b, 0, 0].unsqueeze(0).expand(len(rows), S)
        o32 = ref_attention_n_rows(q[rows], rows, k, v, S, softmax_n_param=n, is_causal=causal, attn_bias=bias, attn_mask=mask,
                                   compute_dtype=torch.float32).float().numpy()
        assert np.abs(o32 - g["o_f32"][hi]).max() <= 1e-3 * max(1e-2, np.abs(g["o_f32"][hi]).max())
        # C oracle on the same rows (bias / mask / causal expressed per row)
        add = np.zeros((len(rows), S), np.float32) if bias is None else bias.float().numpy()
        vis = np.ones((len(rows), S), bool) if mask is None else mask.numpy().copy()
        if causal:
            vis &= np.arange(S)[None, :] <= g["rows"][:, None]
        oc = c_oracle.attention_n(q[rows].float().numpy()[None, None], k.float().numpy()[None, None], v.float().numpy()[None, None],
                                  n=n, mask=vis[None, None], bias=add[None, None])
        assert np.abs(oc[0, 0] - g["o_f32"][hi]).max() <= 2e-5


@pytest.mark.parametrize("cfg", ["c2", "c4"])
def test_g5_backward_rows(golden_dir, cfg):
    g = _g(golden_dir, f"g5_{cfg}.npz")
    dtype = G4[cfg]
    B, H, S, D = (int(x) for x in g["shape"])
    b, h = (int(x) for x in g["head"])
    n, causal = float(g["n"]), bool(g["causal"])
    q, k, v, do = (_head_slice(nm, (B, H, S, D), dtype, b, h) for nm in ("q", "k", "v", "dout"))
    assert [synth.checksum(t) for t in (q, k, v, do)] == list(g["checksums"])
    bias = mask = None
    if cfg == "c4":
        bias = synth.alibi_bias_rows(H, S, S, [h], np.arange(S), dtype)[0].float()
        mask = synth.keypad_mask(B, S)[b, 0, 0].unsqueeze(0).expand(S, S)
    qq, kk, vv = (t.float().requires_grad_() for t in (q, k, v))
    o = ref_attention_n(qq, kk, vv, softmax_n_param=n, is_causal=causal, attn_bias=bias, attn_mask=mask)
    o.backward(do.float())
    rows = g["rows"]
    assert np.abs(o.detach().numpy()[rows] - g["o"]).max() <= 1e-5
    for name, t in (("dq", qq), ("dk", kk), ("dv", vv)):
        assert np.abs(t.grad.numpy()[rows] - g[name]).max() <= 1e-4 * max(1.0, np.abs(g[name]).max())

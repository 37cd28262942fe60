"""Parity tests proper (-m gpu): the HIP path (Python front end -> ctypes -> C ABI -> gfx950 kernels) against the oracle on
the same seeded inputs, against the committed golden fixtures (outputs of the real reference), and — at BASELINE.json's full
sizes — through size-independent properties. Tolerances are the reference's own (tests/gpu/core/test_flash_attn.py:14:
atol 1e-2 fp16 / 5e-2 bf16, rtol 0) plus tighter relative gates against the fp32 "true" answer."""
import os

import numpy as np
import pytest
import torch

from oracle import c_oracle
from oracle.ref_attention import analytic_answer, analytic_causal_answer, ref_attention_n

import flash_attention_softmax_n_amd.synth as synth

pytestmark = pytest.mark.gpu

REF_ATOL = {torch.float16: 1e-2, torch.bfloat16: 5e-2, torch.float32: 1e-3}   # reference GPU test tolerances (rtol 0)
REL_TRUE = {torch.float16: 2.0 ** -9, torch.bfloat16: 2.0 ** -6, torch.float32: 2e-5}  # vs fp32 oracle, relative to the tensor's max |x|


def _rand(shape, dtype, dev, seed, std=0.5):
    return synth.counter_normal(shape, seed, std=std, dtype=dtype, device=dev)


def _check(got, want, dtype, what):
    got, want = got.detach().float().cpu(), want.detach().float().cpu()
    assert torch.isfinite(got).all(), f"{what}: non-finite values"
    err = (got - want).abs().max().item()
    # the reference's absolute tolerance is written for O(1) tensors; a gradient summed over hundreds of rows is not
    atol = REF_ATOL[dtype] * max(1.0, want.abs().max().item())
    assert err <= atol, f"{what}: max-abs {err:.3e} > reference atol {atol:.3e}"
    lim = REL_TRUE[dtype] * max(want.abs().max().item(), 1e-2)
    assert err <= lim, f"{what}: max-abs {err:.3e} > {lim:.3e} (relative gate)"


def _oracle_fwd_bwd(q, k, v, do, **kw):
    qc, kc, vc = (t.detach().cpu().float().requires_grad_() for t in (q, k, v))
    kw = {a: ((b.detach().cpu() if b.is_cuda else b) if torch.is_tensor(b) else b) for a, b in kw.items()}
    o = ref_attention_n(qc, kc, vc, **kw)
    o.backward(do.detach().cpu().float())
    return o, qc.grad, kc.grad, vc.grad


# ---------------------------------------------------------------- paired causal launches
@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("L,S", [(1152, 1152), (1152, 1408), (1408, 1152), (640, 640)])   # (1408, 1152): the first 256 rows see no key at all
def test_causal_launches_that_pair_their_blocks(pkg, dev, L, S, dtype):
    """Causal launches with at least two rounds of workgroups put block r and block nblk-1-r of a head into one workgroup
    (csrc/fasn_launch.h: forward, dQ and dK/dV). Odd block counts (9 and 5 x 128 rows: the middle block runs alone), L != S
    (bottom-right aligned diagonal) and every head compared with the oracle on a few (batch, head) slices, all rows finite."""
    B, H, D = 13, 32, 64   # 416 heads x 9 (5) blocks >= 2 x 768 (512) workgroup slots
    q = _rand((B, H, L, D), dtype, dev, 11).requires_grad_()
    k, v = (_rand((B, H, S, D), dtype, dev, s).requires_grad_() for s in (12, 13))
    do = _rand((B, H, L, D), dtype, dev, 14, std=1.0)
    out = pkg.flash_attention_n(q, k, v, softmax_n_param=1.0, is_causal=True)
    out.backward(do)
    for t in (out, q.grad, k.grad, v.grad):
        assert torch.isfinite(t).all()
    for b, h in ((0, 0), (5, 17), (12, 31)):
        sl = (slice(b, b + 1), slice(h, h + 1))
        o, dq, dk, dv = _oracle_fwd_bwd(q[sl], k[sl], v[sl], do[sl], softmax_n_param=1.0, is_causal=True)
        _check(out[sl], o, dtype, f"out[{b},{h}]")
        _check(q.grad[sl], dq, dtype, f"dq[{b},{h}]")
        _check(k.grad[sl], dk, dtype, f"dk[{b},{h}]")
        _check(v.grad[sl], dv, dtype, f"dv[{b},{h}]")


@pytest.mark.parametrize("D", [32, 64])
@pytest.mark.parametrize("kind", ["bias", "bias_b1ls", "dense", "bias+dense"])
@pytest.mark.parametrize("L,S", [(1152, 1152), (1152, 1408), (1408, 1152), (640, 644)])
def test_causal_launches_with_a_mask_or_bias_pair_their_blocks(pkg, dev, L, S, kind, D):
    """Round 6: the vector mask / bias modes of a CAUSAL call (ALiBi in a decoder) pair their blocks like the plain causal kernels - forward, dQ and
    dK/dV of head dims 32 / 64 (one-wave kernels) - with a batch-broadcast bias (the per-XCD (head, block pair, batch) order), a per-batch bias and
    dense masks; odd block counts, L != S, forward and gradients on a few (batch, head) slices against the oracle, everything finite."""
    dtype = torch.bfloat16
    B, H = 13, 32
    q = _rand((B, H, L, D), dtype, dev, 11).requires_grad_()
    k, v = (_rand((B, H, S, D), dtype, dev, s_).requires_grad_() for s_ in (12, 13))
    do = _rand((B, H, L, D), dtype, dev, 14, std=1.0)
    gen = torch.Generator().manual_seed(4)
    kw = dict(softmax_n_param=1.0, is_causal=True)
    if kind in ("bias", "bias+dense"):
        kw["attn_bias"] = torch.randn(H, L, S, generator=gen).to(dtype).to(dev)
    if kind == "bias_b1ls":
        kw["attn_bias"] = torch.randn(B, 1, L, S, generator=gen).to(dtype).to(dev)
    if "dense" in kind:
        m = torch.rand(B, 1, L, S, generator=gen) < 0.8
        m[..., 0] = True
        kw["attn_mask"] = m.to(dev)
    out = pkg.flash_attention_n(q, k, v, **kw)
    out.backward(do)
    for t in (out, q.grad, k.grad, v.grad):
        assert torch.isfinite(t).all()
    for b, h in ((0, 0), (5, 17), (12, 31)):
        sl = (slice(b, b + 1), slice(h, h + 1))
        okw = dict(kw)
        if "attn_bias" in okw:
            okw["attn_bias"] = (okw["attn_bias"][h:h + 1] if kind != "bias_b1ls" else okw["attn_bias"][b:b + 1]).float()
        if "attn_mask" in okw:
            okw["attn_mask"] = okw["attn_mask"][b:b + 1]
        o, dq, dk, dv = _oracle_fwd_bwd(q[sl], k[sl], v[sl], do[sl], **okw)
        for got, want, nm in ((out[sl], o, "out"), (q.grad[sl], dq, "dq"), (k.grad[sl], dk, "dk"), (v.grad[sl], dv, "dv")):
            _check(got, want, dtype, f"causal {kind} D={D} L{L} S{S} [{b},{h}] {nm}")


@pytest.mark.parametrize("D", [32, 64, 128])   # (128: the forward pairs - 512 blocks of 256 rows on 256 workgroup slots -, the two-wave backward groups its heads)
@pytest.mark.parametrize("kind", ["bias_hls", "bias_b1ls", "dense", "plain", "dropout"])
def test_paired_causal_launches_cover_every_block_once(pkg, dev, kind, D):
    """(4,32,1024,D) causal: 1024 blocks of 128 rows = two rounds - forward, dQ and dK/dV pair their blocks in every mode (plain, dropout, batch-broadcast
    bias = the per-XCD (head, block pair, batch) order, per-batch bias, dense mask). EVERY (batch, head) of forward and gradients against the oracle:
    a block pair mapped twice or not at all shows."""
    dtype = torch.bfloat16
    B, H, L, S = 4, 32, 1024, 1024
    q = _rand((B, H, L, D), dtype, dev, 31).requires_grad_()
    k, v = (_rand((B, H, S, D), dtype, dev, s_).requires_grad_() for s_ in (32, 33))
    do = _rand((B, H, L, D), dtype, dev, 34, std=1.0)
    gen = torch.Generator().manual_seed(6)
    kw = dict(softmax_n_param=1.0, is_causal=True)
    if kind == "bias_hls":
        kw["attn_bias"] = torch.randn(H, L, S, generator=gen).to(dtype).to(dev)
    if kind == "bias_b1ls":
        kw["attn_bias"] = torch.randn(B, 1, L, S, generator=gen).to(dtype).to(dev)
    if kind == "dense":
        m = torch.rand(B, 1, L, S, generator=gen) < 0.8
        m[..., 0] = True
        kw["attn_mask"] = m.to(dev)
    p = 0.1 if kind == "dropout" else 0.0
    torch.manual_seed(5)
    out = pkg.flash_attention_n(q, k, v, dropout_p=p, **kw)
    state = pkg.flash_attn.last_dropout_state() if p > 0 else None
    out.backward(do)
    okw = {a: (c.float() if a == "attn_bias" else c) for a, c in kw.items()}
    if p > 0:
        keep = pkg.dropout.keep_mask(state[0], state[1], B, H, L, S, p)
        o, dq, dk, dv = _oracle_dropout(q, k, v, do, keep, pkg.dropout.effective_p(p), **okw)
    else:
        o, dq, dk, dv = _oracle_fwd_bwd(q, k, v, do, **okw)
    for got, want, nm in ((out, o, "out"), (q.grad, dq, "dq"), (k.grad, dk, "dk"), (v.grad, dv, "dv")):
        _check(got, want, dtype, f"paired causal {kind} D={D} {nm}")


def _torch_reference_on_device(q, k, v, do, n, causal, mask, bias):
    """fp32 attention with softmax_n by torch operators ON THE GPU (gradients by autograd): a completeness check for large grids - was every block
    of every head written, exactly once? - not the parity oracle (that is oracle/, on the CPU, in the tests around this one)."""
    qf, kf, vf = (t.detach().float().requires_grad_() for t in (q, k, v))
    G = qf.shape[1] // kf.shape[1]
    kx, vx = (t.repeat_interleave(G, dim=1) if G > 1 else t for t in (kf, vf))
    L, S = qf.shape[2], kf.shape[2]
    x = qf @ kx.transpose(-1, -2) * qf.shape[-1] ** -0.5
    if bias is not None:
        x = x + bias.float()
    hide = torch.zeros(L, S, dtype=torch.bool, device=q.device)
    if causal:
        hide = torch.arange(S, device=q.device)[None, :] > (torch.arange(L, device=q.device)[:, None] + (S - L))
    if mask is not None:
        hide = hide | ~mask
    x = x.masked_fill(hide, float("-inf"))
    m = x.amax(-1, keepdim=True).clamp_min(0.0) if n > 0 else x.amax(-1, keepdim=True)
    m = torch.where(torch.isfinite(m), m, torch.zeros_like(m))
    e = torch.exp(x - m)
    w = e / (n * torch.exp(-m) + e.sum(-1, keepdim=True)).clamp_min(1e-30)
    o = w @ vx
    o.backward(do.float())
    return o, qf.grad, kf.grad, vf.grad


@pytest.mark.parametrize("seed", range(int(os.environ.get("FASN_FUZZ_GRIDS", "36"))))
def test_randomized_large_causal_grids_write_every_block(pkg, dev, seed):
    """Round 6 changed how causal launches hand their blocks to workgroups (pairs from 1 - 1.25 rounds, head groups, per-XCD orders): random LARGE grids
    - 64 .. 192 heads, 5 .. 12 blocks, every head dim, plain / bias / dense mask / grouped K/V, L != S - with ALL of forward and gradients compared with a
    torch fp32 reference computed on the device (tolerance of a 16-bit kernel against fp32: the point is that no block is missing or written twice;
    FASN_FUZZ_GRIDS=N runs N seeds)."""
    rng = np.random.default_rng(4200 + seed)
    D = int(rng.choice([32, 64, 64, 128, 128, 256]))
    B, H = int(rng.choice([2, 3, 4, 6])), int(rng.choice([16, 32]))
    G = int(rng.choice([1, 1, 2, 4]))
    L = int(rng.choice([640, 768, 1000, 1024, 1152, 1536]))
    S = L if rng.integers(0, 3) else int(L + rng.choice([-128, 64, 200]))
    kind = str(rng.choice(["plain", "plain", "bias_hls", "bias_b1ls", "dense"]))
    dtype = [torch.float16, torch.bfloat16][int(rng.integers(0, 2))]
    q = _rand((B, H, L, D), dtype, dev, 41).requires_grad_()
    k, v = (_rand((B, H // G, S, D), dtype, dev, s_).requires_grad_() for s_ in (42, 43))
    do = _rand((B, H, L, D), dtype, dev, 44, std=1.0)
    gen = torch.Generator().manual_seed(seed)
    mask = bias = None
    if kind == "bias_hls":
        bias = torch.randn(H, L, S, generator=gen).to(dtype).to(dev)
    if kind == "bias_b1ls":
        bias = torch.randn(B, 1, L, S, generator=gen).to(dtype).to(dev)
    if kind == "dense":
        mask = torch.rand(B, 1, L, S, generator=gen) < 0.8
        mask[..., 0] = True
        mask = mask.to(dev)
    out = pkg.flash_attention_n(q, k, v, softmax_n_param=1.0, is_causal=True, attn_mask=mask, attn_bias=bias)
    out.backward(do)
    o, dq, dk, dv = _torch_reference_on_device(q, k, v, do, 1.0, True, mask, bias)
    what = f"D{D} ({B},{H}/{H // G},{L},{S}) {kind} {dtype}"
    for got, want, nm in ((out, o, "out"), (q.grad, dq, "dq"), (k.grad, dk, "dk"), (v.grad, dv, "dv")):
        assert torch.isfinite(got).all(), f"{what} {nm}: non-finite"
        err = (got.float() - want).abs().amax(dim=(-1, -2))          # per (batch, head)
        lim = 0.02 * want.abs().amax(dim=(-1, -2)).clamp_min(0.05)
        bad = (err > lim).nonzero()
        assert bad.numel() == 0, f"{what} {nm}: (batch, head) {bad[:4].tolist()} off by {err.max().item():.3e}"


@pytest.mark.parametrize("D", [128, 256])
@pytest.mark.parametrize("B,H,Hkv,L,S", [(2, 16, 16, 640, 640), (2, 16, 16, 384, 640), (4, 8, 8, 650, 648), (2, 32, 8, 512, 512), (1, 8, 8, 1100, 1100)])
def test_causal_two_wave_kernels_hand_blocks_out_by_head_groups(pkg, dev, B, H, Hkv, L, S, D):
    """Round 6: causal launches of the two-wave kernels (D = 128 dQ / dK/dV, D = 256 forward / dQ / dK/dV) take the heads of an XCD in groups and hand a
    group's blocks out block index by block index (csrc/fasn_common.h: block_to_work_grouped). Batch x heads a multiple of 8 (16, 32), groups of 2 / 4,
    odd block counts, L != S, grouped K/V: EVERY (batch, head) of forward and gradients against the oracle - a block mapped twice or not at all shows."""
    dtype = torch.bfloat16
    q = _rand((B, H, L, D), dtype, dev, 21).requires_grad_()
    k, v = (_rand((B, Hkv, S, D), dtype, dev, s_).requires_grad_() for s_ in (22, 23))
    do = _rand((B, H, L, D), dtype, dev, 24, std=1.0)
    out = pkg.flash_attention_n(q, k, v, softmax_n_param=1.0, is_causal=True)
    out.backward(do)
    G = H // Hkv
    kx, vx = (t.detach().repeat_interleave(G, dim=1) for t in (k, v))
    o, dq, dkx, dvx = _oracle_fwd_bwd(q, kx, vx, do, softmax_n_param=1.0, is_causal=True)
    dk, dv = (t.view(B, Hkv, G, S, D).sum(2) for t in (dkx, dvx))
    for got, want, nm in ((out, o, "out"), (q.grad, dq, "dq"), (k.grad, dk, "dk"), (v.grad, dv, "dv")):
        _check(got, want, dtype, f"causal D={D} ({B},{H}/{Hkv},{L},{S}) {nm}")


@pytest.mark.parametrize("seed", range(8))
def test_folded_causal_kernel_agrees_with_the_32_row_kernel(pkg, dev, seed):
    """Causal launches of 2048+ 256-row blocks take the folded two-phase forward kernel (round 5, FOLD in csrc/fasn_fwd_kernel.h: rows folded
    w / 7 - w, unmasked main walk + per-32-row diagonal walk) and paired pipelined backward kernels; the same inputs one batch element at a
    time stay below the rule and take the 32-rows-per-wave kernel with single blocks. Random ragged L, S (L != S both ways, bottom-right
    aligned diagonal, rows without keys), both dtypes: outputs and gradients of the two routes agree to rounding (same arithmetic, other
    summation order), and a (batch, head) slice agrees with the oracle."""
    rng = np.random.default_rng(7000 + seed)
    dtype = [torch.float16, torch.bfloat16][seed & 1]
    B, H, D = 16, 32, 64
    L = int(rng.integers(1024, 1500))
    S = int(np.clip(L + rng.integers(-300, 300), 1, None))
    n = float(rng.choice([0.0, 1.0, 0.5]))
    if L > S and n == 0.0:
        n = 1.0   # (rows without a visible key: softmax_0 over an empty set is 0/0 in the oracle; the kernels return zeros)
    q = _rand((B, H, L, D), dtype, dev, 21 + seed).requires_grad_()
    k, v = (_rand((B, H, S, D), dtype, dev, s + seed).requires_grad_() for s in (22, 23))
    do = _rand((B, H, L, D), dtype, dev, 24 + seed, std=1.0)
    assert B * H * ((L + 255) // 256) >= 2048
    out = pkg.flash_attention_n(q, k, v, softmax_n_param=n, is_causal=True)
    out.backward(do)
    big = [t.detach().clone() for t in (out, q.grad, k.grad, v.grad)]
    q.grad = k.grad = v.grad = None
    outs = []
    for b in range(B):
        ob = pkg.flash_attention_n(q[b:b + 1], k[b:b + 1], v[b:b + 1], softmax_n_param=n, is_causal=True)
        ob.backward(do[b:b + 1])
        outs.append(ob.detach())
    small = [torch.cat(outs), q.grad, k.grad, v.grad]
    for a, b_, nm in zip(big, small, ("out", "dq", "dk", "dv")):
        assert torch.isfinite(a).all(), nm
        scale = max(1.0, b_.float().abs().max().item())
        err = (a.float() - b_.float()).abs().max().item()
        assert err <= 4 * REL_TRUE[dtype] * scale, f"{nm}: folded vs 32-row route differ by {err:.3e} (L={L}, S={S}, n={n})"
    sl = (slice(3, 4), slice(5, 6))
    o, dq, dk, dv = _oracle_fwd_bwd(q[sl], k[sl], v[sl], do[sl], softmax_n_param=n, is_causal=True)
    for got, want, nm in ((big[0][sl], o, "out"), (big[1][sl], dq, "dq"), (big[2][sl], dk, "dk"), (big[3][sl], dv, "dv")):
        _check(got, want, dtype, f"folded {nm} (L={L}, S={S})")


@pytest.mark.parametrize("causal", [False, True])
@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_dynamic_deal_across_xcds_is_the_static_deal_bit_for_bit(pkg, dev, causal, dtype):
    """Long plain / causal launches at head dim 64 hand fasn_fwd_ws 64 bytes of workspace and deal their (head, query block) items dynamically
    across XCDs (round 5, csrc/fasn_fwd_kernel.h: draw_item; eight counters zeroed by the library, surplus workgroups leave at once). Which
    workgroup runs an item changes nothing about the item: output and LSE must equal the static deal's (fasn_fwd, no workspace) bit for bit,
    every row written exactly once, twice in a row (the counters are zeroed per launch), also with ragged rows."""
    from flash_attention_softmax_n_amd import _lib, flash_attn
    lib = _lib.load()
    B, H, D = 32, 32, 64
    L = 1000 if not causal else 2000    # 4 x 1024 (8 x 1024 causal) blocks of 256 rows: the rule's 8 rounds
    S = L + (24 if causal else 0)
    q = _rand((B, H, L, D), dtype, dev, 31)
    k, v = (_rand((B, H, S, D), dtype, dev, s) for s in (32, 33))
    outs = []
    for use_ws in (False, True, True):
        o = torch.full((B, H, L, D), float("nan"), dtype=dtype, device=dev)
        lse = torch.full((B, H, L), float("nan"), dtype=torch.float32, device=dev)
        a = _lib.FwdArgs()
        flash_attn._fill_fwd(a, q, k, v, o, lse, None, None, 1.0, D ** -0.5, causal, 0.0)
        need = lib.fasn_fwd_workspace_bytes(a)
        assert need == 64
        if use_ws:
            ws = torch.full((need,), 0x5A, dtype=torch.uint8, device=dev)   # (not zero: the library zeroes its counters itself)
            assert lib.fasn_fwd_ws(a, ws.data_ptr(), need, flash_attn._stream_ptr(dev)) == 0
        else:
            assert lib.fasn_fwd(a, flash_attn._stream_ptr(dev)) == 0
        torch.cuda.synchronize()
        assert torch.isfinite(o).all() and torch.isfinite(lse).all()
        outs.append((o, lse))
    for o, lse in outs[1:]:
        assert torch.equal(o, outs[0][0]) and torch.equal(lse, outs[0][1])
    # and the front end takes the workspace route on its own
    got = pkg.flash_attention_n(q, k, v, softmax_n_param=1.0, is_causal=causal)
    assert torch.equal(got, outs[0][0])


# ---------------------------------------------------------------- the backward's scratch
@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("case", ["plain", "causal_rows_without_keys", "ragged", "keypad", "gqa", "bias", "dropout", "d32", "d128"])
def test_backward_scratch_is_written_before_it_is_read(pkg, dev, monkeypatch, case, dtype):
    """delta = rowsum(dO o O) is a scratch buffer of the caller's that fasn_bwd fills itself (fasn_bwd_delta_kernel; the reference's
    _bwd_preprocess, core/flash_attn_triton.py:129-143). The scratch is filled with NaN first: every row the dQ / dK/dV kernels read
    has to have been written (rows without a visible key and rows past a ragged end included - P = 0 there, and 0 x NaN would still
    poison dK), the results have to be bit-identical to a run with a clean scratch and equal to the oracle's gradients."""
    B, H, L, S, D = 2, 4, 200, 333, 64
    kw, Hkv, drop = dict(softmax_n_param=1.0), 4, 0.0
    if case == "causal_rows_without_keys":
        L, S = 333, 200
        kw["is_causal"] = True
    elif case == "plain":
        L, S = 512, 512
    elif case == "keypad":
        m = torch.ones(B, 1, 1, S, dtype=torch.bool, device=dev)
        m[0, ..., 100:] = False
        m[1, ..., 300:] = False
        kw["attn_mask"] = m
    elif case == "gqa":
        Hkv = 2
    elif case == "bias":
        kw["attn_bias"] = (0.5 * torch.randn(1, H, L, S, device=dev)).to(dtype)
    elif case == "dropout":
        drop = 0.2
    elif case == "d32":
        D = 32
    elif case == "d128":
        D = 128
    q = _rand((B, H, L, D), dtype, dev, 21).requires_grad_()
    k, v = (_rand((B, Hkv, S, D), dtype, dev, s).requires_grad_() for s in (22, 23))
    do = _rand((B, H, L, D), dtype, dev, 24, std=1.0)

    def run():
        for t in (q, k, v):
            t.grad = None
        torch.manual_seed(5)
        out = pkg.flash_attention_n(q, k, v, dropout_p=drop, **kw)
        out.backward(do)
        return out.detach(), q.grad.clone(), k.grad.clone(), v.grad.clone()

    clean = run()
    monkeypatch.setattr(pkg.flash_attn, "_POISON_SCRATCH", True)
    poisoned = run()
    for a, b, nm in zip(clean, poisoned, ("out", "dq", "dk", "dv")):
        assert torch.isfinite(b).all(), nm
        assert torch.equal(a, b), nm
    if case not in ("dropout",):
        ke, ve = (t.repeat_interleave(H // Hkv, dim=1) for t in (k, v))
        okw = dict(kw)
        if "attn_mask" in okw:
            okw["attn_mask"] = okw["attn_mask"].cpu()
        if "attn_bias" in okw:
            okw["attn_bias"] = okw["attn_bias"].float().cpu()
        o, dq, dk, dv = _oracle_fwd_bwd(q, ke, ve, do, **okw)
        g = H // Hkv
        dk, dv = (t.reshape(B, Hkv, g, S, D).sum(2) for t in (dk, dv))
        _check(poisoned[0], o, dtype, "out")
        _check(poisoned[1], dq, dtype, "dq")
        _check(poisoned[2], dk, dtype, "dk")
        _check(poisoned[3], dv, dtype, "dv")


# ---------------------------------------------------------------- the reference's own GPU test, same grid
@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("is_causal", [False, True])
@pytest.mark.parametrize("scale", [None, 0.1, 0.5])
@pytest.mark.parametrize("n", [0, 1, 4])
def test_flash_attention_n_vs_oracle(pkg, dev, n, scale, is_causal, dtype):
    """reference tests/gpu/core/test_flash_attn.py:10-48: shape (6,1,1024,64), fwd + dq/dk/dv"""
    shape = (6, 1, 1024, 64)
    q, k, v = (_rand(shape, dtype, dev, s).requires_grad_() for s in (1, 2, 3))
    do = _rand(shape, dtype, dev, 4, std=1.0)
    out = pkg.flash_attention_n(q, k, v, softmax_n_param=n, scale=scale, is_causal=is_causal)
    out.backward(do)
    o, dq, dk, dv = _oracle_fwd_bwd(q, k, v, do, softmax_n_param=float(n), scale=scale, is_causal=is_causal)
    _check(out, o, dtype, "out")
    _check(q.grad, dq, dtype, "dq")
    _check(k.grad, dk, dtype, "dk")
    _check(v.grad, dv, dtype, "dv")
    # the reference test's literal criterion: atol vs slow_attention_n run in the NATIVE dtype, rtol 0, gradients included
    qn, kn, vn = (t.detach().cpu().requires_grad_() for t in (q, k, v))
    on = ref_attention_n(qn, kn, vn, softmax_n_param=float(n), scale=scale, is_causal=is_causal)
    on.backward(do.cpu())
    for what, got, want in (("out", out, on), ("dq", q.grad, qn.grad), ("dk", k.grad, kn.grad), ("dv", v.grad, vn.grad)):
        err = (got.detach().float().cpu() - want.detach().float()).abs().max().item()
        assert err <= REF_ATOL[dtype], f"{what}: max-abs {err:.3e} vs native-dtype oracle > literal atol {REF_ATOL[dtype]}"


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("weight", [10, 3, 0.5, 0.04, 0.02, 0.01, 0, -0.01, -0.02, -0.04, -0.5, -3, -10])
@pytest.mark.parametrize("n", [0, 1, 4])
def test_flash_attention_analytic(pkg, dev, n, weight, dtype):
    """reference tests/gpu/core/test_flash_attn.py:51-91: Q=K=V=w, N=6, L=1024, S=1152 (L != S), scale 0.3, atol 1e-3"""
    N, L, S, E, scale = 6, 1024, 1152, 64, 0.3
    q = weight * torch.ones(N, 1, L, E, device=dev, dtype=dtype)
    k = weight * torch.ones(N, 1, S, E, device=dev, dtype=dtype)
    v = weight * torch.ones(N, 1, S, E, device=dev, dtype=dtype)
    w = float(q[0, 0, 0, 0])  # the weight after rounding to dtype
    a = pkg.flash_attention_n(q, k, v, scale=scale, softmax_n_param=n).float().cpu()
    want = analytic_answer(w, S, E, scale, n)
    assert (a - want).abs().max().item() <= 1e-3 + 2.0 ** -8 * abs(want)
    b = pkg.flash_attention_n(q, k, v, scale=scale, softmax_n_param=n, is_causal=True).float().cpu()
    wantc = torch.tensor(analytic_causal_answer(w, L, S, E, scale, n))
    rtol = 2e-2 if dtype == torch.bfloat16 else 2e-3
    got = b.sum(dim=0).sum(dim=-1)[0]
    assert torch.allclose(got, N * E * wantc, rtol=rtol, atol=1e-3)


# ---------------------------------------------------------------- the reference's Triton test file, same grids (rows a6 / a8 / a9)
_T_SHAPE = (32, 16, 1024, 32)   # reference tests/gpu/core/test_flash_attn_triton.py:21-23: batch_size (32, 16), L = S = 1024, E = 32, fp16
_T_SLICES = [(0, 0), (3, 7), (9, 2), (14, 15), (17, 4), (21, 9), (26, 11), (31, 15)]   # (batch, head) slices the CPU oracle is run on


def _triton_grid_case(pkg, dev, sm_n, scale, is_causal, atol):
    """One case of the reference's Triton grid: flash_attention_n_triton(query, key, value, is_causal, scale, softmax_n_param) at the
    reference's shape and dtype, forward and dq / dk / dv, against slow_attention_n's op sequence in the NATIVE dtype (what the reference test
    compares with) at the reference's literal atol, rtol 0. The kernel runs the whole (32, 16) batch; the CPU oracle a spread of its
    (batch, head) slices (the slices of one launch are independent problems), and every element of the full outputs must be finite."""
    dtype = torch.float16
    q, k, v = (_rand(_T_SHAPE, dtype, dev, s).requires_grad_() for s in (41, 42, 43))
    do = _rand(_T_SHAPE, dtype, dev, 44, std=1.0)   # (randn_like(actual): N(0, 1))
    out = pkg.flash_attention_n_triton(q, k, v, is_causal=is_causal, scale=scale, softmax_n_param=sm_n)
    assert out.dtype == dtype and out.shape == q.shape
    out.backward(do)
    for nm, t in (("out", out), ("dq", q.grad), ("dk", k.grad), ("dv", v.grad)):
        assert torch.isfinite(t).all(), nm
    worst = {}
    for b, h in _T_SLICES:
        sl = (slice(b, b + 1), slice(h, h + 1))
        qn, kn, vn = (t.detach()[sl].cpu().requires_grad_() for t in (q, k, v))
        on = ref_attention_n(qn, kn, vn, softmax_n_param=float(sm_n), scale=scale, is_causal=is_causal)
        on.backward(do[sl].cpu())
        for nm, got, want in (("out", out, on), ("dq", q.grad, qn.grad), ("dk", k.grad, kn.grad), ("dv", v.grad, vn.grad)):
            err = (got.detach()[sl].float().cpu() - want.detach().float()).abs().max().item()
            worst[nm] = max(worst.get(nm, 0.0), err)
    for nm, err in worst.items():
        assert err <= atol, f"{nm}: max-abs {err:.3e} vs native-dtype oracle > the reference's atol {atol} (n={sm_n}, scale={scale}, causal={is_causal})"
    # and the fp32 "true" answer on one slice, relative gate (the literal atol is loose next to outputs of ~0.02)
    sl = (slice(3, 4), slice(7, 8))
    o, dq, dk, dv = _oracle_fwd_bwd(q[sl], k[sl], v[sl], do[sl], softmax_n_param=float(sm_n), scale=scale, is_causal=is_causal)
    for got, want, nm in ((out[sl], o, "out"), (q.grad[sl], dq, "dq"), (k.grad[sl], dk, "dk"), (v.grad[sl], dv, "dv")):
        _check(got, want, dtype, f"triton grid {nm} (n={sm_n}, scale={scale}, causal={is_causal})")


@pytest.mark.parametrize("scale", [None, 0.5, 0.01, 0.4, 0.3, 0.02])
@pytest.mark.parametrize("sm_n", [0., 1., 1e-3, 1e-6, 4., 3.])
def test_triton_signature_reference_grid(pkg, dev, sm_n, scale):
    """reference tests/gpu/core/test_flash_attn_triton.py:13-48: (32,16,1024,32) fp16, n x scale grid, forward + dq / dk / dv, atol 2e-3."""
    _triton_grid_case(pkg, dev, sm_n, scale, False, 2e-3)


@pytest.mark.parametrize("scale", [None, 0.5, 0.01, 0.4, 0.3, 0.02])
@pytest.mark.parametrize("sm_n", [0., 1e-6])
def test_triton_signature_reference_grid_causal(pkg, dev, sm_n, scale):
    """reference tests/gpu/core/test_flash_attn_triton.py:51-86: the causal grid, atol 2e-2."""
    _triton_grid_case(pkg, dev, sm_n, scale, True, 2e-2)


@pytest.mark.parametrize("scale", [None, 0.01, 0.3, 0.02])
def test_triton_signature_reference_grid_causal_1em3(pkg, dev, scale):
    """reference tests/gpu/core/test_flash_attn_triton.py:89-124: causal with n = 1e-3, atol 2e-2."""
    _triton_grid_case(pkg, dev, 1e-3, scale, True, 2e-2)


@pytest.mark.parametrize("weight", [10, 3, 0.5, 0.04, 0.02, 0.01, 0, -0.01, -0.02, -0.04, -0.5, -3, -10])
@pytest.mark.parametrize("sm_n", [0., 1., 1e-3, 1e-6, 4.])
def test_triton_signature_analytic(pkg, dev, sm_n, weight):
    """reference tests/gpu/core/test_flash_attn_triton.py:127-169: Q = K = V = weight, N = 6, L = 1024, S = 1152, E = Ev = 64, scale 0.3, fp16
    through flash_attention_n_triton: the closed form at atol 1e-3 (the reference's figure) without the causal flag, the causal row sums at
    rtol 2e-3, and agreement with slow_attention_n's op sequence in fp16 at the same tolerances."""
    N, L, S, E, scale, dtype = 6, 1024, 1152, 64, 0.3, torch.float16
    q = weight * torch.ones(N, 1, L, E, device=dev, dtype=dtype)
    k = weight * torch.ones(N, 1, S, E, device=dev, dtype=dtype)
    v = weight * torch.ones(N, 1, S, E, device=dev, dtype=dtype)
    w = float(q[0, 0, 0, 0])   # the weight after rounding to fp16 (the reference's closed form uses the unrounded one and allows 1e-3 for it)
    a = pkg.flash_attention_n_triton(q, k, v, scale=scale, softmax_n_param=sm_n)
    assert a.dtype == dtype
    want = analytic_answer(w, S, E, scale, sm_n)
    af = a.float().cpu()
    assert (af - want).abs().max().item() <= 1e-3 + 2.0 ** -10 * abs(want), (sm_n, weight)
    slow = ref_attention_n(q[:1].cpu(), k[:1].cpu(), v[:1].cpu(), softmax_n_param=sm_n, scale=scale).float()
    assert (af[:1] - slow).abs().max().item() <= 1e-3 + 2.0 ** -9 * abs(want), (sm_n, weight)
    b = pkg.flash_attention_n_triton(q, k, v, is_causal=True, scale=scale, softmax_n_param=sm_n).float().cpu()
    wantc = torch.tensor(analytic_causal_answer(w, L, S, E, scale, sm_n))
    got = b.sum(dim=0).sum(dim=-1)[0]
    assert torch.allclose(got, N * E * wantc, rtol=2e-3, atol=1e-6), (sm_n, weight)


# ---------------------------------------------------------------- golden fixtures (outputs of the real reference)
@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("causal", [False, True])
@pytest.mark.parametrize("n", [0.0, 0.5, 1.0, 4.0])
def test_golden_g1(pkg, dev, golden_dir, n, causal, dtype):
    """BASELINE config 1 (2,2,128,32): the inputs are exact in both 16-bit types, so the only error is the kernel's own"""
    g = np.load(os.path.join(golden_dir, "g1_c1.npz"))
    tag = f"n{n}_c{int(causal)}"
    q, k, v = (torch.from_numpy(g[x]).to(dtype).to(dev).requires_grad_() for x in ("q", "k", "v"))
    for x, t in (("q", q), ("k", k), ("v", v)):
        assert torch.equal(t.detach().float().cpu(), torch.from_numpy(g[x]))  # exactly representable
    out = pkg.flash_attention_n(q, k, v, softmax_n_param=n, is_causal=causal)
    out.backward(torch.from_numpy(g["dout"]).to(dtype).to(dev))
    _check(out, torch.from_numpy(g[f"o_{tag}"]), dtype, "o")
    _check(q.grad, torch.from_numpy(g[f"dq_{tag}"]), dtype, "dq")
    _check(k.grad, torch.from_numpy(g[f"dk_{tag}"]), dtype, "dk")
    _check(v.grad, torch.from_numpy(g[f"dv_{tag}"]), dtype, "dv")
    if dtype == torch.bfloat16:  # what the reference's tests compare against: its own bf16-eager output, atol 5e-2
        assert (out.detach().float().cpu() - torch.from_numpy(g[f"o_bf16native_{tag}"])).abs().max().item() <= 5e-2


G4 = {"c2": torch.bfloat16, "c3": torch.float16, "m0": torch.bfloat16, "c4": torch.bfloat16, "c5": torch.bfloat16}
SEEDS = {"q": 101, "k": 102, "v": 103, "dout": 104}


def _full(name, shape, dtype, dev):
    return synth.counter_normal(shape, SEEDS[name], std=1.0 if name == "dout" else 0.5, dtype=dtype, device=dev)


@pytest.mark.parametrize("cfg", ["c2", "c3", "m0", "c4", "c5"])
def test_golden_g4_full_size_sampled_rows(pkg, dev, golden_dir, cfg):
    """BASELINE configs at FULL size; sampled rows of 4 (b,h) heads against the reference's slow_attention_n"""
    g = np.load(os.path.join(golden_dir, f"g4_{cfg}.npz"))
    dtype = G4[cfg]
    B, H, S, D = (int(x) for x in g["shape"])
    n, causal = float(g["n"]), bool(g["causal"])
    q, k, v = (_full(nm, (B, H, S, D), dtype, dev) for nm in ("q", "k", "v"))
    for hi, (b, h) in enumerate(g["heads"]):  # generator produced the fixture's bits on this machine too
        assert [synth.checksum(t[int(b), int(h)]) for t in (q, k, v)] == list(g["checksums"][hi])
    bias = mask = None
    if cfg == "c4":
        bias = synth.alibi_bias(H, S, S, dtype, device=dev)      # dense [H,L,S], as the reference requires
        mask = synth.keypad_mask(B, S, device=dev)               # [B,1,1,S]
    out = pkg.flash_attention_n(q, k, v, softmax_n_param=n, is_causal=causal, attn_mask=mask, attn_bias=bias)
    rows = torch.from_numpy(g["rows"]).to(dev)
    for hi, (b, h) in enumerate(g["heads"]):
        got = out[int(b), int(h)][rows].float().cpu()
        true = torch.from_numpy(g["o_f32"][hi])
        native = torch.from_numpy(g["o_native"][hi])
        assert (got - native).abs().max().item() <= REF_ATOL[dtype]   # the reference-test criterion
        assert (got - native).abs().max().item() < 1e-2              # north_star: max-abs < 1e-2 vs slow_attention_n
        err = (got - true).abs().max().item()
        rms = true.pow(2).mean().sqrt().item()
        gate = (2.0 ** -8 if dtype == torch.bfloat16 else 2.0 ** -10) * max(true.abs().max().item(), 4 * rms)
        assert err <= gate, f"{cfg} head {hi}: max-abs {err:.3e} vs fp32 oracle > {gate:.3e}"


@pytest.mark.parametrize("cfg", ["c2", "c3", "m0", "c4"])
def test_golden_g5_backward_rows(pkg, dev, golden_dir, cfg):
    g = np.load(os.path.join(golden_dir, f"g5_{cfg}.npz"))
    dtype = G4[cfg]
    B, H, S, D = (int(x) for x in g["shape"])
    b, h = (int(x) for x in g["head"])
    n, causal = float(g["n"]), bool(g["causal"])
    start = ((b * H + h) * S) * D
    q, k, v, do = (synth.counter_normal((1, 1, S, D), SEEDS[nm], std=1.0 if nm == "dout" else 0.5, dtype=dtype, device=dev, start=start)
                   for nm in ("q", "k", "v", "dout"))
    assert [synth.checksum(t) for t in (q, k, v, do)] == list(g["checksums"])
    bias = mask = None
    if cfg == "c4":
        bias = synth.alibi_bias_rows(H, S, S, [h], np.arange(S), dtype).to(dev)   # [1,S,S]
        mask = synth.keypad_mask(B, S, device=dev)[b:b + 1]
    q.requires_grad_(), k.requires_grad_(), v.requires_grad_()
    out = pkg.flash_attention_n(q, k, v, softmax_n_param=n, is_causal=causal, attn_mask=mask, attn_bias=bias)
    out.backward(do)
    rows = torch.from_numpy(g["rows"]).to(dev)
    _check(out[0, 0][rows], torch.from_numpy(g["o"]), dtype, "o")
    _check(q.grad[0, 0][rows], torch.from_numpy(g["dq"]), dtype, "dq")
    _check(k.grad[0, 0][rows], torch.from_numpy(g["dk"]), dtype, "dk")
    _check(v.grad[0, 0][rows], torch.from_numpy(g["dv"]), dtype, "dv")


@pytest.mark.parametrize("cfg", ["c2", "c3", "m0", "c5", "c4"])
def test_golden_g5f_backward_at_full_grid(pkg, dev, golden_dir, cfg):
    """Forward + backward of every BASELINE config at its FULL (B,H,S,D) grid (C4: dense ALiBi [H,L,S] + key padding, S = 8192,
    D = 128; C5: B = 64); sampled rows of O, dQ, dK, dV of three (b,h) heads against the real reference's slow_attention_n +
    autograd. Two gates per tensor: the reference test's literal criterion (tests/gpu/core/test_flash_attn.py:14,46-48: atol
    1e-2 fp16 / 5e-2 bf16, rtol 0, against slow_attention_n in the NATIVE dtype) and a relative gate against the fp32 answer."""
    g = np.load(os.path.join(golden_dir, f"g5f_{cfg}.npz"))
    dtype = G4[cfg]
    B, H, S, D = (int(x) for x in g["shape"])
    n, causal = float(g["n"]), bool(g["causal"])
    q, k, v, do = (_full(nm, (B, H, S, D), dtype, dev) for nm in ("q", "k", "v", "dout"))
    for hi, (b, h) in enumerate(g["heads"]):
        assert [synth.checksum(t[int(b), int(h)]) for t in (q, k, v, do)] == list(g["checksums"][hi])
    bias = mask = None
    if cfg == "c4":
        bias = synth.alibi_bias(H, S, S, dtype, device=dev)
        mask = synth.keypad_mask(B, S, device=dev)
    q.requires_grad_(), k.requires_grad_(), v.requires_grad_()
    out = pkg.flash_attention_n(q, k, v, softmax_n_param=n, is_causal=causal, attn_mask=mask, attn_bias=bias)
    out.backward(do)
    rows = torch.from_numpy(g["rows"]).to(dev)
    for hi, (b, h) in enumerate(g["heads"]):
        for name, t in (("o", out), ("dq", q.grad), ("dk", k.grad), ("dv", v.grad)):
            got = t[int(b), int(h)][rows].detach().float().cpu()
            true, native = torch.from_numpy(g[name][hi]), torch.from_numpy(g[name + "_native"][hi])
            assert torch.isfinite(got).all()
            lit = (got - native).abs().max().item()
            assert lit <= REF_ATOL[dtype], f"{cfg} head {hi} {name}: max-abs {lit:.3e} vs native slow_attention_n > atol {REF_ATOL[dtype]}"
            err = (got - true).abs().max().item()
            lim = REL_TRUE[dtype] * max(true.abs().max().item(), 1e-2)
            assert err <= lim, f"{cfg} head {hi} {name}: max-abs {err:.3e} vs fp32 slow_attention_n > {lim:.3e}"


# ---------------------------------------------------------------- backward plan
def test_backward_needs_no_workspace_and_ignores_the_reserved_flag(pkg, dev):
    """libfasn.so has ONE backward plan (dQ kernel + dK/dV kernel, deterministic): fasn_bwd_workspace_bytes is 0 whatever `flags` says
    (the one-pass 5-GEMM kernel lives in the developer library only since round 4) and two runs agree bit for bit"""
    from flash_attention_softmax_n_amd import _lib
    from flash_attention_softmax_n_amd.flash_attn import _fill_fwd, _view4
    lib = _lib.load()
    q = torch.zeros(1, 2, 64, 64, dtype=torch.bfloat16, device=dev)
    lse = torch.zeros(1, 2, 64, dtype=torch.float32, device=dev)
    a = _lib.BwdArgs()
    _fill_fwd(a.fwd, q, q, q, q, lse, None, None, 1.0, 0.125, False)
    a.dout = a.dq = a.dk = a.dv = _view4(q)
    for flags in (0, _lib.FASN_BWD_ONE_PASS):
        a.flags = flags
        assert lib.fasn_bwd_workspace_bytes(a) == 0
    qq, kk, vv = (_rand((2, 4, 1100, 64), torch.bfloat16, dev, s).requires_grad_() for s in (1, 2, 3))
    do = _rand((2, 4, 1100, 64), torch.bfloat16, dev, 4, std=1.0)
    grads = []
    for _ in range(2):
        qq.grad = kk.grad = vv.grad = None
        pkg.flash_attention_n(qq, kk, vv, softmax_n_param=1.0, is_causal=True).backward(do)
        grads.append([g.clone() for g in (qq.grad, kk.grad, vv.grad)])
    for g0, g1 in zip(*grads):
        assert torch.equal(g0, g1)


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("causal", [False, True])
@pytest.mark.parametrize("L,S", [(1024, 1024), (100, 77), (3, 5), (300, 200), (1100, 1300), (1300, 1100), (33, 1000), (513, 1537), (64, 1), (1, 64), (129, 127)])
def test_pipelined_backward_ragged_sizes(pkg, dev, L, S, causal, dtype):
    """the software-pipelined D = 64 dQ / dK/dV kernels (csrc/fasn_bwd_pipe.h: plain and causal launches): ragged tiles, L != S
    (bottom-right causal alignment), query tiles that see no key, one-block launches, n in {0, 1}"""
    n = 1.0 if (L % 2 or L > S) else 0.0   # (n = 0 with fully hidden rows - causal L > S - is 0/0 in the oracle's autograd)
    B, H, D = 2, 3, 64
    q, k, v = (_rand(sh, dtype, dev, s).requires_grad_() for sh, s in (((B, H, L, D), 1), ((B, H, S, D), 2), ((B, H, S, D), 3)))
    do = _rand((B, H, L, D), dtype, dev, 4, std=1.0)
    out = pkg.flash_attention_n(q, k, v, softmax_n_param=n, is_causal=causal)
    out.backward(do)
    _, dq, dk, dv = _oracle_fwd_bwd(q, k, v, do, softmax_n_param=n, is_causal=causal)
    for got, want, nm in ((q.grad, dq, "dq"), (k.grad, dk, "dk"), (v.grad, dv, "dv")):
        _check(got, want, dtype, f"{nm} L{L} S{S} causal{causal}")


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("kind", ["plain", "causal", "bias", "keypad+causal"])
@pytest.mark.parametrize("shape", [(2, 8, 1024, 1024, 256), (1, 2, 200, 333, 160), (2, 2, 77, 100, 192), (1, 1, 3, 5, 256)])
def test_head_dim_256(pkg, dev, shape, kind, dtype):
    """core/flash_attn.py:117-124 takes any head dim; here D in (128, 256] runs the D = 256 kernels (zero-padded features), forward
    and backward, against the oracle: (2,8,1024,256) plain / causal / ALiBi bias, ragged sizes, L != S."""
    B, H, L, S, D = shape
    n = 0.5
    q = _rand((B, H, L, D), dtype, dev, 31).requires_grad_()
    k, v = (_rand((B, H, S, D), dtype, dev, s_).requires_grad_() for s_ in (32, 33))
    do = _rand((B, H, L, D), dtype, dev, 34, std=1.0)
    kw = {}
    if kind == "bias":
        kw["attn_bias"] = synth.alibi_bias(H, L, S, dtype, device=dev)
    if kind in ("causal", "keypad+causal"):
        kw["is_causal"] = True
    if kind == "keypad+causal":
        kw["attn_mask"] = synth.keypad_mask(B, S, device=dev)
        n = 1.0
    out = pkg.flash_attention_n(q, k, v, softmax_n_param=n, **kw)
    out.backward(do)
    o, dq, dk, dv = _oracle_fwd_bwd(q, k, v, do, softmax_n_param=n, **kw)
    for got, want, nm in ((out, o, "out"), (q.grad, dq, "dq"), (k.grad, dk, "dk"), (v.grad, dv, "dv")):
        _check(got, want, dtype, f"D={D} {kind} {nm}")


def test_head_dim_limits(pkg, dev):
    """above 256 (16-bit) / 128 (fp32) the call is refused with the reason, not served by a fallback"""
    q = torch.zeros(1, 1, 8, 264, dtype=torch.bfloat16, device=dev)
    with pytest.raises(NotImplementedError):
        pkg.flash_attention_n(q, q, q)
    q32 = torch.zeros(1, 1, 8, 160, dtype=torch.float32, device=dev)
    with pytest.raises(NotImplementedError):
        pkg.flash_attention_n(q32, q32, q32)


# ---------------------------------------------------------------- masks, bias, layouts, edge cases
@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("D", [32, 64, 128, 256])
@pytest.mark.parametrize("kind", ["keypad", "dense", "bias3d", "bias4d_f32", "all"])
def test_mask_bias_combinations(pkg, dev, kind, D, dtype):
    B, H, L, S = 2, 3, 200, 264
    q, k, v = (_rand(sh, dtype, dev, s).requires_grad_() for sh, s in (((B, H, L, D), 1), ((B, H, S, D), 2), ((B, H, S, D), 3)))
    do = _rand((B, H, L, D), dtype, dev, 4, std=1.0)
    mask = bias = None
    causal = kind == "all"
    gen = torch.Generator().manual_seed(5)
    if kind in ("keypad", "all"):
        mask = synth.keypad_mask(B, S, device=dev)
    if kind == "dense":
        mask = (torch.rand(B, H, L, S, generator=gen) < 0.7).to(dev)
        mask[..., 0] = True
    if kind in ("bias3d", "all"):
        bias = torch.randn(H, L, S, generator=gen).to(dtype).to(dev)
    if kind == "bias4d_f32":
        bias = torch.randn(B, 1, L, S, generator=gen).to(dev)
    out = pkg.flash_attention_n(q, k, v, softmax_n_param=0.5, attn_mask=mask, attn_bias=bias, is_causal=causal)
    out.backward(do)
    o, dq, dk, dv = _oracle_fwd_bwd(q, k, v, do, softmax_n_param=0.5, attn_mask=mask, is_causal=causal,
                                    attn_bias=None if bias is None else bias.float())
    for got, want, nm in ((out, o, "out"), (q.grad, dq, "dq"), (k.grad, dk, "dk"), (v.grad, dv, "dv")):
        _check(got, want, dtype, f"{kind}/{nm}")


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("D", [32, 64, 128])
@pytest.mark.parametrize("with_mask", [False, True, "dense"])
def test_fp32_bias_next_to_16_bit_inputs(pkg, dev, with_mask, D, dtype):
    """An fp32 additive bias next to fp16 / bf16 q / k / v (what Hugging Face models hand over; the reference adds the mask in whatever dtype
    SDPA is given, core/flash_attn.py:100-113) on the vector path (round 5): fp32 images in the forward (D = 128: the 8-wave register-staged
    kernel) and the one-wave dQ kernel, the fp32 bias ring of the two-wave dK/dV kernel at D = 128 (one-wave elsewhere). Several key blocks and
    row blocks, ragged ends, a head-broadcast [H,L,S] ALiBi with values a 16-bit bias could not hold next to a key-padding mask."""
    B, H, L, S = 3, 4, 300, 432
    q, k, v = (_rand(sh, dtype, dev, s).requires_grad_() for sh, s in (((B, H, L, D), 1), ((B, H, S, D), 2), ((B, H, S, D), 3)))
    do = _rand((B, H, L, D), dtype, dev, 4, std=1.0)
    bias = synth.alibi_bias(H, L, S, torch.float32, device=dev) + 1e-3 * torch.randn(H, L, S, generator=torch.Generator().manual_seed(9)).to(dev)
    mask = synth.keypad_mask(B, S, device=dev) if with_mask else None
    if with_mask == "dense":   # a per-(batch, row) boolean mask with aligned rows: the mask image next to the fp32 bias image
        mask = (torch.rand(B, 1, L, S, generator=torch.Generator().manual_seed(5)) < 0.8).to(dev)
        mask[..., 0] = True
    from flash_attention_softmax_n_amd.flash_attn import kernel_path
    assert kernel_path(q, k, v, attn_mask=mask, attn_bias=bias) in ("vector mask/bias", "vector bias + key-padding")
    out = pkg.flash_attention_n(q, k, v, softmax_n_param=0.5, attn_mask=mask, attn_bias=bias)
    out.backward(do)
    o, dq, dk, dv = _oracle_fwd_bwd(q, k, v, do, softmax_n_param=0.5, attn_mask=mask, attn_bias=bias)
    for got, want, nm in ((out, o, "out"), (q.grad, dq, "dq"), (k.grad, dk, "dk"), (v.grad, dv, "dv")):
        _check(got, want, dtype, f"fp32 bias D={D} mask={with_mask} {nm}")


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("D", [32, 64, 128])
@pytest.mark.parametrize("why", ["mask_rows_not_dword_movable", "mask_pointer_misaligned"])
def test_fp32_bias_with_a_mask_that_forces_element_loads(pkg, dev, why, D, dtype):
    """An aligned fp32 bias ([1,1,1,S]: vector-movable on its own) next to a dense boolean mask whose rows are NOT 4-byte movable (Sk % 4 != 0,
    or a mask view that starts at an odd byte): the call as a whole is MODE_GENERAL_SLOW. The forward always took the element-load kernel for it;
    the backward of round 5 looked at the bias alone and launched the fp32-image vector kernels, which DMA mask rows from misaligned offsets
    (csrc/fasn_bwd_launch.h). Forward and every gradient against the oracle, and the path query says element loads."""
    B, H, L = 2, 3, 200
    S = 331 if why == "mask_rows_not_dword_movable" else 332
    q, k, v = (_rand(sh, dtype, dev, s).requires_grad_() for sh, s in (((B, H, L, D), 1), ((B, H, S, D), 2), ((B, H, S, D), 3)))
    do = _rand((B, H, L, D), dtype, dev, 4, std=1.0)
    bias = (-0.01 * torch.arange(S, dtype=torch.float32, device=dev)).view(1, 1, 1, S) + 0.25
    gen = torch.Generator().manual_seed(7)
    if why == "mask_rows_not_dword_movable":
        mask = (torch.rand(B, H, L, S, generator=gen) < 0.8).to(dev)
    else:
        buf = (torch.rand(B * H * L * S + 8, generator=gen) < 0.8).to(dev)
        mask = buf[1:1 + B * H * L * S].view(B, H, L, S)   # rows of 332 bytes starting at an odd address
        assert mask.data_ptr() % 4 != 0
    mask[..., 0] = True
    from flash_attention_softmax_n_amd.flash_attn import kernel_path
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore", RuntimeWarning)   # (the front end says "element loads" once per kind of call)
        assert kernel_path(q, k, v, attn_mask=mask, attn_bias=bias) == "element-load (slow)"
        out = pkg.flash_attention_n(q, k, v, softmax_n_param=0.5, attn_mask=mask, attn_bias=bias)
        out.backward(do)
    o, dq, dk, dv = _oracle_fwd_bwd(q, k, v, do, softmax_n_param=0.5, attn_mask=mask, attn_bias=bias)
    for got, want, nm in ((out, o, "out"), (q.grad, dq, "dq"), (k.grad, dk, "dk"), (v.grad, dv, "dv")):
        _check(got, want, dtype, f"fp32 bias + {why} D={D} {nm}")


@pytest.mark.parametrize("D", [32, 64, 128])
def test_fp32_bias_with_an_fp16_scale_that_forbids_prescaled_operands(pkg, dev, D):
    """fp16 q / k / v with |scale * log2e| > 8 (scale 6: the pre-scaled operand could leave the fp16 range) and an aligned fp32 bias: the forward
    takes the element-load kernel, which scales in fp32, and so must the backward - round 5's backward launched the fp32-image vector kernels,
    which pre-scale K in fp16. Inputs small enough that every score is moderate; gradients against the oracle."""
    dtype = torch.float16
    B, H, L, S = 2, 2, 160, 264
    q = (_rand((B, H, L, D), dtype, dev, 1).float() * 0.25).to(dtype).requires_grad_()
    k = (_rand((B, H, S, D), dtype, dev, 2).float() * 0.25).to(dtype).requires_grad_()
    v = _rand((B, H, S, D), dtype, dev, 3).requires_grad_()
    do = _rand((B, H, L, D), dtype, dev, 4, std=1.0)
    bias = synth.alibi_bias(H, L, S, torch.float32, device=dev)
    import warnings
    from flash_attention_softmax_n_amd.flash_attn import kernel_path
    with warnings.catch_warnings():
        warnings.simplefilter("ignore", RuntimeWarning)
        assert kernel_path(q, k, v, attn_bias=bias, scale=6.0) == "element-load (slow)"
        out = pkg.flash_attention_n(q, k, v, softmax_n_param=1.0, scale=6.0, attn_bias=bias)
        out.backward(do)
    o, dq, dk, dv = _oracle_fwd_bwd(q, k, v, do, softmax_n_param=1.0, scale=6.0, attn_bias=bias)
    for got, want, nm in ((out, o, "out"), (q.grad, dq, "dq"), (k.grad, dk, "dk"), (v.grad, dv, "dv")):
        _check(got, want, dtype, f"fp32 bias, fp16 scale 6, D={D} {nm}")


@pytest.mark.parametrize("S", [257, 262, 263])
def test_mask_bias_views_with_odd_key_tails(pkg, dev, S):
    """mask / bias handed over as views into wider buffers: rows stay aligned (vector loads) while the key count leaves a
    partial dword at the end of every row - the last keys must still be read"""
    dtype = torch.bfloat16
    B, H, L, D, W = 2, 2, 130, 64, 272
    q, k, v = (_rand(sh, dtype, dev, s).requires_grad_() for sh, s in (((B, H, L, D), 1), ((B, H, S, D), 2), ((B, H, S, D), 3)))
    do = _rand((B, H, L, D), dtype, dev, 4, std=1.0)
    gen = torch.Generator().manual_seed(11)
    bias = (2.0 * torch.randn(1, H, L, W, generator=gen)).to(dtype).to(dev)[..., :S]
    mask = (torch.rand(B, 1, L, W, generator=gen) < 0.6).to(dev)[..., :S]
    mask[..., S - 1] = True          # the very last key is visible and carries weight
    bias[..., S - 1] += 3.0
    assert bias.stride(2) == W and mask.stride(2) == W
    for m, bb in ((mask, bias), (mask, None), (None, bias)):
        for t in (q, k, v):
            t.grad = None
        out = pkg.flash_attention_n(q, k, v, softmax_n_param=1, attn_mask=m, attn_bias=bb)
        out.backward(do)
        o, dq, dk, dv = _oracle_fwd_bwd(q, k, v, do, softmax_n_param=1.0, attn_mask=m,
                                        attn_bias=None if bb is None else bb.float())
        for got, want, nm in ((out, o, "out"), (q.grad, dq, "dq"), (k.grad, dk, "dk"), (v.grad, dv, "dv")):
            _check(got, want, dtype, f"S={S} {nm}")


@pytest.mark.parametrize("D", [32, 64, 128])
@pytest.mark.parametrize("L,S", [(1, 4096), (5, 5000), (64, 2048), (130, 4100)])
@pytest.mark.parametrize("kind", ["plain", "causal", "bias+mask", "mask"])
def test_decode_shapes_take_the_split_key_path(pkg, dev, kind, L, S, D):
    """few query rows, many keys: the keys of one (b,h) are split over workgroups and merged by the
    combine kernel; same answers as the oracle, sink (+n) counted once, backward unchanged"""
    from flash_attention_softmax_n_amd import _lib as L_
    from flash_attention_softmax_n_amd.flash_attn import _fill_fwd
    dtype = torch.bfloat16
    B, H = 2, 3
    q, k, v = (_rand(sh, dtype, dev, s).requires_grad_() for sh, s in (((B, H, L, D), 1), ((B, H, S, D), 2), ((B, H, S, D), 3)))
    do = _rand((B, H, L, D), dtype, dev, 4, std=1.0)
    mask = bias = None
    if kind == "bias+mask":
        gen = torch.Generator().manual_seed(3)
        mask = synth.keypad_mask(B, S, device=dev)
        bias = torch.randn(1, H, L, S, generator=gen).to(dtype).to(dev)
    if kind == "mask":      # key-padding mask alone: the forward's key-padding mode hands over to the split-K mask kernel
        mask = synth.keypad_mask(B, S, device=dev)
    # the plan really is split-K for these shapes
    a = L_.FwdArgs()
    o_ = torch.empty_like(q)
    lse_ = torch.empty((B, H, L), dtype=torch.float32, device=dev)
    _fill_fwd(a, q.detach(), k.detach(), v.detach(), o_, lse_, None if mask is None else mask.expand(B, H, L, S).view(torch.uint8),
              None if bias is None else bias.expand(B, H, L, S), 0.5, D ** -0.5, kind == "causal")
    if not (kind == "mask" and S % 4):      # (an unaligned lone key mask has no vector form: single-pass kernel)
        assert L_.load().fasn_fwd_workspace_bytes(a) > 0
    out = pkg.flash_attention_n(q, k, v, softmax_n_param=0.5, is_causal=kind == "causal", attn_mask=mask, attn_bias=bias)
    out.backward(do)
    o, dq, dk, dv = _oracle_fwd_bwd(q, k, v, do, softmax_n_param=0.5, is_causal=kind == "causal", attn_mask=mask,
                                    attn_bias=None if bias is None else bias.float())
    for got, want, nm in ((out, o, "out"), (q.grad, dq, "dq"), (k.grad, dk, "dk"), (v.grad, dv, "dv")):
        _check(got, want, dtype, f"{kind} L={L} S={S} D={D} {nm}")


@pytest.mark.parametrize("causal", [False, True])
@pytest.mark.parametrize("L", [2048, 2000])
def test_head_dim_128_large_grid_sampled_rows(pkg, dev, L, causal):
    """(2,32,L,128): enough 256-row blocks that the D=128 forward takes its 8-waves-per-workgroup kernel (full and ragged last
    block, key count not a tile multiple); sampled rows of three heads against the oracle"""
    from oracle.ref_attention import ref_attention_n_rows
    dtype = torch.bfloat16
    B, H, D, S = 2, 32, 128, L + 48
    q, k, v = (_rand(sh, dtype, dev, s) for sh, s in (((B, H, L, D), 1), ((B, H, S, D), 2), ((B, H, S, D), 3)))
    out = pkg.flash_attention_n(q, k, v, softmax_n_param=1, is_causal=causal)
    rows = torch.tensor([0, 1, 31, 32, 255, 256, 257, 1000, L - 1])
    for b, h in ((0, 0), (1, 31), (0, 17)):
        want = ref_attention_n_rows(q[b, h][rows.to(dev)].float().cpu(), rows, k[b, h].float().cpu(), v[b, h].float().cpu(), L,
                                    softmax_n_param=1.0, is_causal=causal)
        _check(out[b, h][rows.to(dev)], want, dtype, f"head ({b},{h})")


@pytest.mark.parametrize("shape", [(2, 1, 3, 8), (1, 2, 1, 64), (1, 1, 65, 16), (3, 2, 127, 96), (1, 1, 257, 40)])
@pytest.mark.parametrize("causal", [False, True])
def test_ragged_and_padded_feature_dims(pkg, dev, shape, causal):
    """tiny / ragged lengths and head dims that are zero-padded to a kernel size; (2,1,3,8) is the reference CPU test shape
    (tests/cpu/core/test_flash_attn.py:12-13)"""
    dtype = torch.bfloat16
    B, H, L, E = shape
    S, Ev = L + 5, E
    q, k, v = (_rand(sh, dtype, dev, s).requires_grad_() for sh, s in (((B, H, L, E), 1), ((B, H, S, E), 2), ((B, H, S, Ev), 3)))
    do = _rand((B, H, L, Ev), dtype, dev, 4, std=1.0)
    out = pkg.flash_attention_n(q, k, v, softmax_n_param=1, is_causal=causal)
    assert out.shape == (B, H, L, Ev)
    out.backward(do)
    o, dq, dk, dv = _oracle_fwd_bwd(q, k, v, do, softmax_n_param=1.0, is_causal=causal)
    for got, want, nm in ((out, o, "out"), (q.grad, dq, "dq"), (k.grad, dk, "dk"), (v.grad, dv, "dv")):
        _check(got, want, dtype, nm)


def test_value_dim_differs_and_shared_kv(pkg, dev):
    """Ev != E (reference README.md:50) and 3-D key/value shared by all heads (flash_attn.py:75-79)"""
    dtype = torch.float16
    B, H, L, S, E, Ev = 2, 4, 100, 130, 64, 32
    q = _rand((B, H, L, E), dtype, dev, 1).requires_grad_()
    k = _rand((B, S, E), dtype, dev, 2).requires_grad_()
    v = _rand((B, S, Ev), dtype, dev, 3).requires_grad_()
    do = _rand((B, H, L, Ev), dtype, dev, 4, std=1.0)
    out = pkg.flash_attention_n(q, k, v, softmax_n_param=2)
    out.backward(do)
    qc, kc, vc = (t.detach().cpu().float().requires_grad_() for t in (q, k, v))
    o = ref_attention_n(qc, kc.unsqueeze(1), vc.unsqueeze(1), softmax_n_param=2.0)
    o.backward(do.cpu().float())
    for got, want, nm in ((out, o, "out"), (q.grad, qc.grad, "dq"), (k.grad, kc.grad, "dk"), (v.grad, vc.grad, "dv")):
        _check(got, want, dtype, nm)


def test_strided_layout_without_copy(pkg, dev):
    """[B, L, H, D] memory viewed as [B, H, L, D] goes to the kernel through strides"""
    dtype = torch.bfloat16
    B, H, L, D = 2, 4, 192, 64
    qkv = _rand((B, L, 3, H, D), dtype, dev, 9)
    q, k, v = (qkv[:, :, i].permute(0, 2, 1, 3) for i in range(3))
    assert not q.is_contiguous()
    out = pkg.flash_attention_n(q, k, v, softmax_n_param=1, is_causal=True)
    o = ref_attention_n(q.cpu().float(), k.cpu().float(), v.cpu().float(), softmax_n_param=1.0, is_causal=True)
    _check(out, o, dtype, "out")


def test_fully_hidden_rows_give_zero(pkg, dev):
    dtype = torch.float16
    q, k, v = (_rand((1, 2, 64, 64), dtype, dev, s) for s in (1, 2, 3))
    mask = torch.ones(1, 1, 64, 64, dtype=torch.bool, device=dev)
    mask[:, :, 10] = False
    out = pkg.flash_attention_n(q, k, v, softmax_n_param=0, attn_mask=mask)
    assert torch.isfinite(out).all() and out[:, :, 10].abs().max().item() == 0.0
    # Sq > Sk causal: the first Sq - Sk rows see nothing
    out = pkg.flash_attention_n(q, k[:, :, :40], v[:, :, :40], softmax_n_param=0, is_causal=True)
    assert out[:, :, :24].abs().max().item() == 0.0 and torch.isfinite(out).all()


def test_real_valued_n_and_signature_aliases(pkg, dev):
    dtype = torch.bfloat16
    q, k, v = (_rand((2, 2, 128, 64), dtype, dev, s) for s in (1, 2, 3))
    for n in (1e-6, 1e-3, 0.5, 3.0):
        a = pkg.flash_attention_n(q, k, v, softmax_n_param=n)
        b = pkg.flash_attention_n_triton(q, k, v, False, None, n)
        c = pkg.slow_attention_n(q, k, v, softmax_n_param=n)
        assert torch.equal(a, b) and torch.equal(a, c)
        _check(a, ref_attention_n(q.cpu().float(), k.cpu().float(), v.cpu().float(), softmax_n_param=n), dtype, f"n={n}")
    # softmax_dtype (functional.py:72-73,91): value's dtype is the only one the reference's matmul accepts; the weights reach P.V rounded to it,
    # as in the kernel - same result as the default, within the reference's bf16 atol of the native-dtype op sequence
    e = pkg.slow_attention_n(q, k, v, softmax_n_param=1.0, softmax_dtype=dtype)
    assert torch.equal(e, pkg.slow_attention_n(q, k, v, softmax_n_param=1.0))
    native = ref_attention_n(q.cpu(), k.cpu(), v.cpu(), softmax_n_param=1.0).float()
    assert (e.float().cpu() - native).abs().max().item() <= REF_ATOL[dtype]
    with pytest.raises(RuntimeError, match="same dtype"):
        pkg.slow_attention_n(q, k, v, softmax_n_param=1.0, softmax_dtype=torch.float32)
    fm = torch.randn(128, 128).to(dev)
    d = pkg.slow_attention_n(q[0], k[0], v[0], attn_mask=fm, softmax_n_param=1.0)   # 3-D inputs + (L,S) float mask
    _check(d, ref_attention_n(q[0].cpu().float(), k[0].cpu().float(), v[0].cpu().float(), softmax_n_param=1.0, attn_bias=fm.cpu()),
           dtype, "slow float mask")
    with pytest.raises(ValueError):
        pkg.flash_attention_n(q, k, v, dropout_p=1.0)
    with pytest.raises(NotImplementedError):
        pkg.flash_attention_n(q.double(), k.double(), v.double())


# ---------------------------------------------------------------- fp32 (exact-fp32 MFMA kernels)
@pytest.mark.parametrize("is_causal", [False, True])
@pytest.mark.parametrize("scale", [None, 0.1, 0.5])
@pytest.mark.parametrize("n", [0, 1, 4])
def test_flash_attention_n_fp32_reference_grid(pkg, dev, n, scale, is_causal):
    """reference tests/gpu/core/test_flash_attn.py:10-48 for float32: shape (6,1,1024,64), atol 1e-3, fwd + grads"""
    shape = (6, 1, 1024, 64)
    q, k, v = (_rand(shape, torch.float32, dev, s).requires_grad_() for s in (1, 2, 3))
    do = _rand(shape, torch.float32, dev, 4, std=1.0)
    out = pkg.flash_attention_n(q, k, v, softmax_n_param=n, scale=scale, is_causal=is_causal)
    assert out.dtype == torch.float32
    out.backward(do)
    o, dq, dk, dv = _oracle_fwd_bwd(q, k, v, do, softmax_n_param=float(n), scale=scale, is_causal=is_causal)
    for got, want, nm in ((out, o, "out"), (q.grad, dq, "dq"), (k.grad, dk, "dk"), (v.grad, dv, "dv")):
        err = (got.detach().cpu() - want).abs().max().item()
        assert err <= 1e-3, f"{nm}: {err:.3e} > reference atol 1e-3"
        assert err <= 2e-5 * max(want.abs().max().item(), 1.0), f"{nm}: {err:.3e} (fp32 MFMA should be near exact)"


@pytest.mark.parametrize("shape", [(2, 2, 100, 32), (1, 3, 257, 128), (2, 1, 3, 8), (1, 2, 130, 96)])
@pytest.mark.parametrize("causal", [False, True])
def test_fp32_ragged_sizes_and_golden(pkg, dev, golden_dir, shape, causal):
    B, H, L, E = shape
    S = L + 7
    q, k, v = (_rand(sh, torch.float32, dev, s).requires_grad_() for sh, s in (((B, H, L, E), 1), ((B, H, S, E), 2), ((B, H, S, E), 3)))
    do = _rand((B, H, L, E), torch.float32, dev, 4, std=1.0)
    out = pkg.flash_attention_n(q, k, v, softmax_n_param=0.5, is_causal=causal)
    out.backward(do)
    o, dq, dk, dv = _oracle_fwd_bwd(q, k, v, do, softmax_n_param=0.5, is_causal=causal)
    for got, want, nm in ((out, o, "out"), (q.grad, dq, "dq"), (k.grad, dk, "dk"), (v.grad, dv, "dv")):
        assert (got.detach().cpu() - want).abs().max().item() <= 2e-5 * max(want.abs().max().item(), 1.0), nm
    # BASELINE config 1 in fp32 against the reference's own output (golden g1), n = 1
    g = np.load(os.path.join(golden_dir, "g1_c1.npz"))
    tag = f"n1.0_c{int(causal)}"
    gq, gk, gv = (torch.from_numpy(g[x]).to(dev).requires_grad_() for x in ("q", "k", "v"))
    go = pkg.flash_attention_n(gq, gk, gv, softmax_n_param=1, is_causal=causal)
    go.backward(torch.from_numpy(g["dout"]).to(dev))
    for got, name in ((go, "o"), (gq.grad, "dq"), (gk.grad, "dk"), (gv.grad, "dv")):
        assert np.abs(got.detach().cpu().numpy() - g[f"{name}_{tag}"]).max() <= 5e-6


@pytest.mark.parametrize("D", [32, 64, 128])
@pytest.mark.parametrize("kind", ["keypad", "dense", "bias3d", "all", "all+dropout"])
def test_fp32_mask_bias_dropout(pkg, dev, kind, D):
    """fp32 inputs with attn_mask / attn_bias / dropout (the reference's fp32 grid includes dropout_p = 0.2,
    tests/gpu/core/test_flash_attn.py:11-15): exact-fp32 kernels, element-load general instantiation"""
    B, H, L, S = 2, 2, 150, 200
    q, k, v = (_rand(sh, torch.float32, dev, s).requires_grad_() for sh, s in (((B, H, L, D), 1), ((B, H, S, D), 2), ((B, H, S, D), 3)))
    do = _rand((B, H, L, D), torch.float32, dev, 4, std=1.0)
    gen = torch.Generator().manual_seed(5)
    mask = bias = None
    causal = kind.startswith("all")
    if kind in ("keypad", "all", "all+dropout"):
        mask = synth.keypad_mask(B, S, device=dev)
    if kind == "dense":
        mask = (torch.rand(B, H, L, S, generator=gen) < 0.7).to(dev)
        mask[..., 0] = True
    if kind in ("bias3d", "all", "all+dropout"):
        bias = torch.randn(H, L, S, generator=gen).to(dev)
    p = 0.2 if kind == "all+dropout" else 0.0
    torch.manual_seed(77)
    out = pkg.flash_attention_n(q, k, v, softmax_n_param=0.5, attn_mask=mask, attn_bias=bias, is_causal=causal, dropout_p=p)
    assert out.dtype == torch.float32
    out.backward(do)
    if p:
        keep = pkg.dropout.keep_mask(*pkg.flash_attn.last_dropout_state(), B, H, L, S, p)
        o, dq, dk, dv = _oracle_dropout(q, k, v, do, keep, pkg.dropout.effective_p(p), softmax_n_param=0.5, is_causal=causal,
                                        attn_mask=mask.cpu(), attn_bias=bias.cpu())
    else:
        o, dq, dk, dv = _oracle_fwd_bwd(q, k, v, do, softmax_n_param=0.5, attn_mask=mask, is_causal=causal, attn_bias=bias)
    for got, want, nm in ((out, o, "out"), (q.grad, dq, "dq"), (k.grad, dk, "dk"), (v.grad, dv, "dv")):
        err = (got.detach().cpu() - want).abs().max().item()
        assert err <= 1e-3, f"{kind}/{nm}: {err:.3e} > reference atol 1e-3"
        assert err <= 5e-5 * max(want.abs().max().item(), 1.0), f"{kind}/{nm}: {err:.3e}"


# ---------------------------------------------------------------- dropout
def _oracle_dropout(q, k, v, do, keep, p_eff, **kw):
    """attention with an EXPLICIT keep mask on the softmax_n weights (reference functional.py:92: dropout after softmax_n)"""
    from oracle.ref_attention import additive_term, ref_softmax_n
    qc, kc, vc = (t.detach().cpu().float().requires_grad_() for t in (q, k, v))
    L, S = qc.shape[-2], kc.shape[-2]
    scale = kw.get("scale") or qc.shape[-1] ** -0.5
    add = additive_term(L, S, mask=kw.get("attn_mask"), bias=kw.get("attn_bias"), causal=kw.get("is_causal", False),
                        dtype=torch.float32, device="cpu", batch_shape=tuple(qc.shape[:-2]))
    w = qc @ kc.transpose(-2, -1) * scale
    if add is not None:
        w = w + add
    w = ref_softmax_n(w, n=kw.get("softmax_n_param"))
    w = w * torch.from_numpy(keep).float() / (1.0 - p_eff)
    o = w @ vc
    o.backward(do.detach().cpu().float())
    return o, qc.grad, kc.grad, vc.grad


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("mode", ["plain", "causal", "bias+mask", "keypad", "bias", "keypad+causal"])
@pytest.mark.parametrize("D", [32, 64, 128, 256])   # (256, round 6: the vector general kernels with dropout instead of the element-load kernels)
def test_dropout_matches_oracle_with_explicit_mask(pkg, dev, D, mode, dtype):
    """reference: dropout(softmax_n(...)) @ v (functional.py:91-93, flash_attn.py:122). The kernels' keep bits are a pure
    function of (seed, b, h, row, key), mirrored on the host by dropout.keep_mask, so parity is exact up to rounding."""
    B, H, L, S, p = 2, 3, 200, 264, 0.2
    q, k, v = (_rand(sh, dtype, dev, s).requires_grad_() for sh, s in (((B, H, L, D), 1), ((B, H, S, D), 2), ((B, H, S, D), 3)))
    do = _rand((B, H, L, D), dtype, dev, 4, std=1.0)
    kw = {"softmax_n_param": 1.0}
    mask = bias = None
    if "causal" in mode:
        kw["is_causal"] = True
    if "bias" in mode:   # (D = 128: the two-wave backward kernels and the visibility-word forward have dropout instantiations since round 4)
        gen = torch.Generator().manual_seed(3)
        bias = torch.randn(H, L, S, generator=gen).to(dtype).to(dev)
    if "mask" in mode or "keypad" in mode:
        mask = synth.keypad_mask(B, S, device=dev)
    torch.manual_seed(1234)
    out = pkg.flash_attention_n(q, k, v, dropout_p=p, attn_mask=mask, attn_bias=bias, **kw)
    seed, offset = pkg.flash_attn.last_dropout_state()
    out.backward(do)
    keep = pkg.dropout.keep_mask(seed, offset, B, H, L, S, p)
    p_eff = pkg.dropout.effective_p(p)
    assert abs(p_eff - p) < 1.6e-5            # 16-bit threshold: the requested probability is honoured to 2^-16
    assert abs((1.0 - keep.mean()) - p_eff) < 0.01
    o, dq, dk, dv = _oracle_dropout(q, k, v, do, keep, p_eff, attn_mask=None if mask is None else mask.cpu(),
                                    attn_bias=None if bias is None else bias.float().cpu(), **kw)
    for got, want, nm in ((out, o, "out"), (q.grad, dq, "dq"), (k.grad, dk, "dk"), (v.grad, dv, "dv")):
        _check(got, want, dtype, f"dropout/{mode}/{nm}")
    # same seed -> same bits; different seed -> different output; p = 0 path untouched
    torch.manual_seed(1234)
    assert torch.equal(out, pkg.flash_attention_n(q, k, v, dropout_p=p, attn_mask=mask, attn_bias=bias, **kw))
    assert not torch.equal(out, pkg.flash_attention_n(q, k, v, dropout_p=p, attn_mask=mask, attn_bias=bias, **kw))


def _plan_args(pkg, q, k, v, dropout_p=0.0, softmax_n_param=1.0, is_causal=False, attn_mask=None, attn_bias=None):
    """a BwdArgs block for fasn_launch_plan (nothing is launched: output / gradient pointers only have to be aligned device addresses)"""
    a = pkg._lib.BwdArgs()
    B, H, L, D = q.shape
    m8 = None if attn_mask is None else attn_mask.expand(B, H, L, k.shape[2]).view(torch.uint8)
    b4 = None if attn_bias is None else attn_bias.detach().expand(B, H, L, k.shape[2])
    pkg.flash_attn._fill_fwd(a.fwd, q.detach(), k.detach(), v.detach(), q.detach(), torch.empty(B, H, L, device=q.device), m8, b4,
                             softmax_n_param, D ** -0.5, is_causal, dropout_p)
    a.dout, a.dq, a.dk, a.dv = (pkg.flash_attn._view4(t.detach()) for t in (q, q, k, k))
    a.delta = a.fwd.lse
    return a


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("mode", ["plain", "causal", "keypad"])
def test_dropout_at_the_plain_kernels_tuning_points(pkg, dev, mode, dtype):
    """Round 6 (dropout stream definition 2): the keep bits are applied to the PACKED weights behind the row sums, so the dropout forward runs at
    the plain kernels' tuning points - 64 rows per wave with packed row sums for grids of a full round (here (8,16,1024,64): 512 blocks of 256
    rows), three waves per SIMD below - and the backward on the pipelined kernels. Exact parity with the oracle under the explicit mask of the
    host mirror on a spread of (batch, head) slices, forward and gradients; every element of the full outputs finite; the dropped fraction is
    the requested one."""
    B, H, L, D, p = 8, 16, 1024, 64, 0.15
    q, k, v = (_rand((B, H, L, D), dtype, dev, s).requires_grad_() for s in (61, 62, 63))
    do = _rand((B, H, L, D), dtype, dev, 64, std=1.0)
    kw = dict(softmax_n_param=1.0)
    if mode == "causal":
        kw["is_causal"] = True
    if mode == "keypad":
        kw["attn_mask"] = synth.keypad_mask(B, L, device=dev)
    names = [n for n, *_ in pkg._lib.launch_plan_described(_plan_args(pkg, q, k, v, dropout_p=p, **kw), pkg._lib.FASN_PLAN_FWD)]
    assert len(names) == 1 and "DROP=1" in names[0] and "SEED=2" in names[0] and ("QB=2" in names[0]) == (mode != "causal"), names
    torch.manual_seed(123)
    out = pkg.flash_attention_n(q, k, v, dropout_p=p, **kw)
    out.backward(do)
    for t in (out, q.grad, k.grad, v.grad):
        assert torch.isfinite(t).all()
    seed, offset = pkg.flash_attn.last_dropout_state()
    keep = pkg.dropout.keep_mask(seed, offset, B, H, L, L, p)
    assert abs((1.0 - keep.mean()) - pkg.dropout.effective_p(p)) < 1e-3
    for b, h in ((0, 0), (3, 9), (7, 15)):
        sl = (slice(b, b + 1), slice(h, h + 1))
        okw = {a: (c[b:b + 1].cpu() if a == "attn_mask" else c) for a, c in kw.items()}
        o, dq, dk, dv = _oracle_dropout(q[sl], k[sl], v[sl], do[sl], keep[b:b + 1, h:h + 1], pkg.dropout.effective_p(p), **okw)
        for got, want, nm in ((out[sl], o, "out"), (q.grad[sl], dq, "dq"), (k.grad[sl], dk, "dk"), (v.grad[sl], dv, "dv")):
            _check(got, want, dtype, f"dropout {mode} [{b},{h}] {nm}")


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("kind", ["bias", "dense", "bias+dense", "bias+keypad", "bias+causal", "bias+causal_one_round", "dense_ragged"])
def test_head_dim_64_mask_bias_modes_on_large_grids(pkg, dev, kind, dtype):
    """Round 6: the vector mask / bias forward at head dim 64 takes 8 waves x 64 rows with the K/V ring (the plain kernel's shape) once the grid
    holds 256 blocks of 512 rows (bias + key padding, whose length pairing halves the workgroups: 512); below that the 4-wave 32-row kernel.
    (8,16,1024,64) / (8,32,1024,64): the plan names the kernel; forward and gradients against the oracle on a spread of (batch, head) slices
    (`dense_ragged`: L = 1000, S = 1012 - ragged last blocks of the 512-row workgroups), every element of the full tensors finite."""
    # (a causal call's workgroups are unequal: the 8-wave kernel from TWO rounds of 512-row blocks, below that 4 waves x 32 rows - both with paired blocks)
    B, H, L, S, D = (8, 32, 1024, 1024, 64) if kind in ("bias+keypad", "bias+causal") else (8, 16, 1000, 1012, 64) if kind == "dense_ragged" else (8, 16, 1024, 1024, 64)
    q = _rand((B, H, L, D), dtype, dev, 71).requires_grad_()
    k, v = (_rand((B, H, S, D), dtype, dev, s_).requires_grad_() for s_ in (72, 73))
    do = _rand((B, H, L, D), dtype, dev, 74, std=1.0)
    kw = dict(softmax_n_param=1.0)
    gen = torch.Generator().manual_seed(9)
    if "bias" in kind:
        kw["attn_bias"] = torch.randn(H, L, S, generator=gen).to(dtype).to(dev)
    if "dense" in kind:
        m = torch.rand(B, 1, L, S, generator=gen) < 0.8
        m[..., 0] = True
        kw["attn_mask"] = m.to(dev)
    if "keypad" in kind:
        kw["attn_mask"] = synth.keypad_mask(B, S, device=dev)
    if "causal" in kind:
        kw["is_causal"] = True
    names = [n for n, *_ in pkg._lib.launch_plan_described(_plan_args(pkg, q, k, v, **kw), pkg._lib.FASN_PLAN_FWD)]
    if kind == "bias+causal_one_round":
        assert len(names) == 1 and "NW=4" in names[0] and "QB=1" in names[0], names
    else:
        assert len(names) == 1 and "NW=8" in names[0] and "QB=2" in names[0] and "RING=2" in names[0], names
    out = pkg.flash_attention_n(q, k, v, **kw)
    out.backward(do)
    for t in (out, q.grad, k.grad, v.grad):
        assert torch.isfinite(t).all()
    for b, h in ((0, 0), (3, 9), (B - 1, H - 1)):
        sl = (slice(b, b + 1), slice(h, h + 1))
        okw = dict(kw)
        if "attn_bias" in okw:
            okw["attn_bias"] = okw["attn_bias"][h:h + 1].float()
        if "attn_mask" in okw:
            okw["attn_mask"] = okw["attn_mask"][b:b + 1]
        o, dq, dk, dv = _oracle_fwd_bwd(q[sl], k[sl], v[sl], do[sl], **okw)
        for got, want, nm in ((out[sl], o, "out"), (q.grad[sl], dq, "dq"), (k.grad[sl], dk, "dk"), (v.grad[sl], dv, "dv")):
            _check(got, want, dtype, f"D=64 large grid {kind} [{b},{h}] {nm}")


@pytest.mark.parametrize("seed", range(int(os.environ.get("FASN_FUZZ_D256", "48"))))   # FASN_FUZZ_D256=N: N seeds
def test_randomized_head_dim_256_mask_bias_dropout(pkg, dev, seed):
    """Round 6: head dim 256 (and padded 160 / 192) with dense masks, biases, key padding, dropout, grouped K/V, causal, L != S, ragged sizes -
    the two-wave forward / dQ kernels with per-wave images, the one-wave vector dK/dV kernel with one additive tile, their dropout
    instantiations. Forward and dq / dk / dv against the oracle (dropout: under the explicit mask of the host mirror)."""
    rng = np.random.default_rng(9100 + seed)
    D = int(rng.choice([256, 256, 256, 192, 160]))
    B, Hkv, G = int(rng.integers(1, 3)), int(rng.choice([1, 2, 3])), int(rng.choice([1, 1, 2, 4]))
    H = Hkv * G
    L = int(rng.choice([1, 33, 64, 129, 200, 384, 520]))
    S = int(rng.choice([4, 36, 64, 100, 256, 332, 516, 776]))
    dtype = [torch.float16, torch.bfloat16][int(rng.integers(0, 2))]
    causal = bool(rng.integers(0, 2))
    p = float(rng.choice([0.0, 0.0, 0.1, 0.3]))
    mask_kind = str(rng.choice(["none", "dense", "dense_b1", "keypad"]))
    bias_kind = str(rng.choice(["none", "hls", "1hls", "b1ls"]))
    n = float(rng.choice([0.0, 0.5, 1.0]))
    if causal and L > S and n == 0.0:   # rows above the bottom-right aligned diagonal see no key: without a sink the oracle's answer is 0 / 0
        n = 1.0                        # (the kernels return 0 there, a deliberate deviation tested elsewhere)
    q = _rand((B, H, L, D), dtype, dev, 1).requires_grad_()
    k, v = (_rand((B, Hkv, S, D), dtype, dev, s_).requires_grad_() for s_ in (2, 3))
    do = _rand((B, H, L, D), dtype, dev, 4, std=1.0)
    gen = torch.Generator().manual_seed(seed)
    mask = bias = None
    if mask_kind.startswith("dense"):
        mask = torch.rand((B, 1 if mask_kind == "dense_b1" else H, L, S), generator=gen) < 0.75
        mask[..., 0] = True
        mask = mask.to(dev)
    elif mask_kind == "keypad":
        mask = torch.ones(B, 1, 1, S, dtype=torch.bool)
        for b in range(B):
            mask[b, ..., int(torch.randint(1, S + 1, (1,), generator=gen)):] = False
        mask = mask.to(dev)
    if bias_kind != "none":
        shape = {"hls": (H, L, S), "1hls": (1, H, L, S), "b1ls": (B, 1, L, S)}[bias_kind]
        bias = torch.randn(*shape, generator=gen).to(dtype).to(dev)
    what = f"D{D} B{B} H{H}/{Hkv} L{L} S{S} {dtype} causal={causal} p={p} mask={mask_kind} bias={bias_kind} n={n}"
    torch.manual_seed(77 + seed)
    out = pkg.flash_attention_n(q, k, v, softmax_n_param=n, dropout_p=p, attn_mask=mask, attn_bias=bias, is_causal=causal)
    state = pkg.flash_attn.last_dropout_state() if p > 0 else None
    out.backward(do)
    kx, vx = (t.detach().repeat_interleave(G, dim=1) for t in (k, v))   # the oracle sees one K/V head per query head
    okw = dict(softmax_n_param=n, is_causal=causal, attn_mask=None if mask is None else mask.cpu(), attn_bias=None if bias is None else bias.float().cpu())
    if p > 0:
        keep = pkg.dropout.keep_mask(state[0], state[1], B, H, L, S, p)
        o, dq, dkx, dvx = _oracle_dropout(q, kx, vx, do, keep, pkg.dropout.effective_p(p), **okw)
    else:
        o, dq, dkx, dvx = _oracle_fwd_bwd(q, kx, vx, do, **okw)
    dk, dv = (t.view(B, Hkv, G, S, D).sum(2) for t in (dkx, dvx))
    for got, want, nm in ((out, o, "out"), (q.grad, dq, "dq"), (k.grad, dk, "dk"), (v.grad, dv, "dv")):
        _check(got, want, dtype, f"{what} {nm}")


def test_dropout_reference_grid_is_finite_and_unbiased(pkg, dev):
    """the reference's own dropout check (tests/gpu/core/test_flash_attn.py:26-27,41-44) only asks for finite sums; also
    check E[dropout(w)] = w: averaging over seeds approaches the no-dropout output"""
    dtype = torch.bfloat16
    shape = (6, 1, 1024, 64)
    q, k, v = (_rand(shape, dtype, dev, s).requires_grad_() for s in (1, 2, 3))
    out = pkg.flash_attention_n(q, k, v, softmax_n_param=1, dropout_p=0.2, is_causal=True)
    out.backward(_rand(shape, dtype, dev, 4, std=1.0))
    for t in (out, q.grad, k.grad, v.grad):
        assert torch.isfinite(t).all() and isinstance(t.float().sum().item(), float)
    base = pkg.flash_attention_n(q, k, v, softmax_n_param=1).float()
    acc = torch.zeros_like(base)
    for i in range(16):
        acc += pkg.flash_attention_n(q, k, v, softmax_n_param=1, dropout_p=0.2).float()
    assert (acc / 16 - base).abs().max().item() < 0.25 * base.abs().max().item() + 0.01


def test_dropout_stream_follows_the_torch_generator_and_resamples_in_graph_replays(pkg, dev):
    """An eager call's (seed, offset) is the state of torch's CUDA generator at call time (reference: torch's philox state behind
    core/flash_attn.py:122): re-seeding reproduces the sequence, consecutive calls differ; a captured forward + backward draws from
    the device-resident graph stream and resamples on every replay - each replay still matching the host mirror."""
    dtype = torch.bfloat16
    B, H, L, S, D, p = 2, 2, 128, 192, 64, 0.1
    q, k, v = (_rand(sh, dtype, dev, s).requires_grad_() for sh, s in (((B, H, L, D), 1), ((B, H, S, D), 2), ((B, H, S, D), 3)))
    do = _rand((B, H, L, D), dtype, dev, 4, std=1.0)
    torch.manual_seed(99)
    a1 = pkg.flash_attention_n(q, k, v, softmax_n_param=1.0, dropout_p=p)
    s1 = pkg.flash_attn.last_dropout_state()
    a2 = pkg.flash_attention_n(q, k, v, softmax_n_param=1.0, dropout_p=p)
    s2 = pkg.flash_attn.last_dropout_state()
    assert s1[0] == 99 and s2[0] == 99 and s2[1] == s1[1] + 4 and not torch.equal(a1, a2)   # torch's generator advanced by 4 per call
    torch.manual_seed(99)
    b1 = pkg.flash_attention_n(q, k, v, softmax_n_param=1.0, dropout_p=p)
    assert pkg.flash_attn.last_dropout_state() == s1 and torch.equal(a1, b1)
    torch.manual_seed(100)
    c1 = pkg.flash_attention_n(q, k, v, softmax_n_param=1.0, dropout_p=p)
    assert pkg.flash_attn.last_dropout_state()[0] == 100 and not torch.equal(a1, c1)

    # graph: capture forward + backward once, replay three times
    sq, sk, sv = (t.detach().clone().requires_grad_() for t in (q, k, v))
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(2):   # warm-up outside the capture (allocations, device random state)
            o = pkg.flash_attention_n(sq, sk, sv, softmax_n_param=1.0, dropout_p=p)
            o.backward(do)
            sq.grad = sk.grad = sv.grad = None
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        go = pkg.flash_attention_n(sq, sk, sv, softmax_n_param=1.0, dropout_p=p)
        state_t = pkg.flash_attn.last_rng_state()
        go.backward(do)
    outs, states = [], []
    for _ in range(3):
        g.replay()
        torch.cuda.synchronize()
        outs.append((go.detach().clone(), sq.grad.detach().clone(), sk.grad.detach().clone(), sv.grad.detach().clone()))
        states.append(tuple(int(x) & 0xFFFFFFFFFFFFFFFF for x in state_t.cpu().tolist()))
    assert states[1][1] == states[0][1] + 1 and states[2][1] == states[1][1] + 1      # the offset advances on the device
    assert states[0][1] >> 62 == 1                                                     # graph stream: disjoint from the eager one
    assert not torch.equal(outs[0][0], outs[1][0]) and not torch.equal(outs[1][0], outs[2][0])
    for (o_, dq_, dk_, dv_), (seed, offset) in zip(outs, states):
        keep = pkg.dropout.keep_mask(seed, offset, B, H, L, S, p)
        o, dq, dk, dv = _oracle_dropout(q, k, v, do, keep, pkg.dropout.effective_p(p), softmax_n_param=1.0)
        for got, want, nm in ((o_, o, "out"), (dq_, dq, "dq"), (dk_, dk, "dk"), (dv_, dv, "dv")):
            _check(got, want, dtype, f"graph replay/{nm}")


@pytest.mark.parametrize("reentrant", [False, True])
def test_dropout_under_activation_checkpointing(pkg, dev, reentrant):
    """torch.utils.checkpoint restores the CUDA generator state before it recomputes a block: the recomputed forward (and the
    backward that follows it) must draw the masks of the original forward. Two attention layers with dropout, each
    checkpointed, against the same two layers run plainly from the same seed."""
    from torch.utils.checkpoint import checkpoint
    dtype, p = torch.bfloat16, 0.2
    B, H, L, D = 2, 2, 256, 64
    x0 = _rand((B, H, L, D), dtype, dev, 5)
    w = [(_rand((D, D), dtype, dev, 20 + i, std=0.2)) for i in range(4)]

    def layer(x, wq, wk):
        return x + pkg.flash_attention_n(x @ wq, x @ wk, x, softmax_n_param=1.0, dropout_p=p, is_causal=True)

    def run(use_ckpt):
        torch.manual_seed(1234)
        x = x0.clone().requires_grad_()
        ws = [t.clone().requires_grad_() for t in w]
        pkg.flash_attention_n(x0, x0, x0, dropout_p=p)   # a call before the layers: the generator is not at its initial offset
        h = x
        for i in range(2):
            h = checkpoint(layer, h, ws[2 * i], ws[2 * i + 1], use_reentrant=reentrant) if use_ckpt else layer(h, ws[2 * i], ws[2 * i + 1])
        h.float().square().sum().backward()
        return [h.detach()] + [t.grad for t in [x] + ws]

    plain, ckpt = run(False), run(True)
    for a, b_ in zip(plain, ckpt):
        assert torch.equal(a, b_)


# ---------------------------------------------------------------- size-independent properties at full BASELINE sizes
@pytest.mark.parametrize("cfg", ["m0", "c3"])
def test_properties_at_full_size(pkg, dev, cfg):
    dtype = G4[cfg]
    B, H, S, D = 8, 16, 4096, 64
    causal = cfg == "c3"
    q, k, v = (_full(nm, (B, H, S, D), dtype, dev) for nm in ("q", "k", "v"))
    fa = pkg.flash_attn
    o1 = pkg.flash_attention_n(q, k, v, softmax_n_param=1.0, is_causal=causal)
    assert torch.isfinite(o1).all()
    # (1) determinism / idempotence: same launch, same bits
    assert torch.equal(o1, pkg.flash_attention_n(q, k, v, softmax_n_param=1.0, is_causal=causal))
    # (2) linearity in V: O(2V) == 2 O(V) exactly (power-of-two scaling commutes with every rounding)
    o2 = pkg.flash_attention_n(q, k, 2 * v, softmax_n_param=1.0, is_causal=causal)
    if dtype == torch.bfloat16:
        assert torch.equal(o2, 2 * o1)
    else:  # fp16 subnormal outputs (< 6e-5) may round one 2^-24 step differently
        assert (o2.float() - 2 * o1.float()).abs().max().item() <= 2.0 ** -23
    # (3) softmax_n vs softmax_0:  O_n = O_0 * exp(LSE_0 - LSE_n),  exp(LSE_n) = n + exp(LSE_0)
    def fwd_lse(n):
        o = torch.empty_like(q)
        lse = torch.empty(B, H, S, dtype=torch.float32, device=dev)
        a = pkg._lib.FwdArgs()
        fa._fill_fwd(a, q, k, v, o, lse, None, None, n, 1.0 / D ** 0.5, causal)
        pkg._lib.check(pkg._lib.load().fasn_fwd(a, torch.cuda.current_stream().cuda_stream), "fasn_fwd")
        return o, lse
    o0, lse0 = fwd_lse(0.0)
    on, lsen = fwd_lse(1.0)
    # the row sum is taken over the weights AS ROUNDED for the PV product (2^-9 / 2^-12 relative each) and relative to a
    # different running max in the two launches: the identity holds to one operand ulp per row and far better on average
    rel = (torch.exp(lsen.double()) / (1.0 + torch.exp(lse0.double())) - 1.0).abs()
    ulp = 2.0 ** -8 if dtype == torch.bfloat16 else 2.0 ** -11
    assert rel.max().item() <= ulp and rel.mean().item() <= ulp / 16
    pred = o0.float() * torch.exp(lse0 - lsen).unsqueeze(-1)
    tol = (2.0 ** -7 if dtype == torch.bfloat16 else 2.0 ** -10) * o0.float().abs().max().item()
    assert (on.float() - pred).abs().max().item() <= tol
    if not causal:
        # (4) key-permutation invariance (non-causal): permuting K and V rows together changes only summation order
        perm = torch.randperm(S, device=dev)
        op = pkg.flash_attention_n(q, k[:, :, perm], v[:, :, perm], softmax_n_param=1.0)
        assert (op.float() - o1.float()).abs().max().item() <= tol
    else:
        # (4') causality: the first 1024 rows do not depend on later keys
        oc = pkg.flash_attention_n(q[:, :, :1024], k[:, :, :1024], v[:, :, :1024], softmax_n_param=1.0, is_causal=True)
        assert (oc.float() - o1[:, :, :1024].float()).abs().max().item() <= tol


def test_c_oracle_cross_check_midsize(pkg, dev):
    """independent fp64-accumulating C oracle at (2,4,512,64) with n=0.5, causal"""
    dtype = torch.bfloat16
    q, k, v = (_rand((2, 4, 512, 64), dtype, dev, s) for s in (1, 2, 3))
    out = pkg.flash_attention_n(q, k, v, softmax_n_param=0.5, is_causal=True).float().cpu().numpy()
    ref = c_oracle.attention_n(q.float().cpu().numpy(), k.float().cpu().numpy(), v.float().cpu().numpy(), n=0.5, causal=True)
    assert np.abs(out - ref).max() <= 2.0 ** -7 * np.abs(ref).max()


# ---------------------------------------------------------------- stand-alone softmax_n kernel
@pytest.mark.parametrize("dtype", [torch.float32, torch.float16, torch.bfloat16])
@pytest.mark.parametrize("n", [0.0, 1.0, 1e-3, 4.0])
def test_softmax_n_kernel(pkg, dev, golden_dir, n, dtype):
    g = np.load(os.path.join(golden_dir, "g3_softmax.npz"))
    if dtype == torch.float32:
        y = pkg.softmax_n(torch.from_numpy(g["x"]).to(dev), n=n)
        assert np.allclose(y.cpu().numpy(), g[f"y_n{n}"], rtol=1e-5, atol=1e-7)
        big = pkg.softmax_n(torch.from_numpy(g["big"]).to(dev), n)
        assert abs(big.sum().item() - 1.0) <= 1e-6
    x = synth.counter_normal((7, 33, 1000), 3, std=2.0, dtype=dtype, device=dev).requires_grad_()
    y = pkg.softmax_n(x, n=n, dim=-1)
    dy = synth.counter_normal((7, 33, 1000), 4, std=1.0, dtype=dtype, device=dev)
    y.backward(dy)
    xc = x.detach().cpu().float().requires_grad_()
    from oracle.ref_attention import ref_softmax_n
    yc = ref_softmax_n(xc, n=n)
    yc.backward(dy.cpu().float())
    tol = {torch.float32: 5e-6, torch.float16: 1e-3, torch.bfloat16: 8e-3}[dtype]
    assert (y.detach().cpu().float() - yc.detach()).abs().max().item() <= tol * max(yc.abs().max().item(), 1e-3) + 1e-7
    assert (x.grad.cpu().float() - xc.grad).abs().max().item() <= 4 * tol * max(xc.grad.abs().max().item(), 1e-3) + 1e-7
    z = pkg.softmax_n(x.detach().transpose(1, 2), n=n, dim=1)  # non-last dim
    assert torch.allclose(z.transpose(1, 2).float(), y.detach().float(), atol=1e-6)
    w = synth.counter_normal((3, 5000), 5, std=3.0, dtype=dtype, device=dev)  # cols > register cache
    assert torch.allclose(pkg.softmax_n(w, n=n).float().cpu(), ref_softmax_n(w.cpu().float(), n=n), atol=tol, rtol=tol)
    # rows too long for one wave's registers: one workgroup per row, forward and backward (12288 and 32768 take the 16-byte kernels,
    # 40000 the element-load kernels)
    for cols in (12288, 32768, 40000):
        xl = synth.counter_normal((5, cols), 6, std=2.0, dtype=dtype, device=dev).requires_grad_()
        dl = synth.counter_normal((5, cols), 7, std=1.0, dtype=dtype, device=dev)
        yl = pkg.softmax_n(xl, n=n)
        yl.backward(dl)
        xr = xl.detach().cpu().float().requires_grad_()
        yr = ref_softmax_n(xr, n=n)
        yr.backward(dl.cpu().float())
        assert (yl.detach().cpu().float() - yr.detach()).abs().max().item() <= tol * max(yr.abs().max().item(), 1e-3) + 1e-7, cols
        assert (xl.grad.cpu().float() - xr.grad).abs().max().item() <= 4 * tol * max(xr.grad.abs().max().item(), 1e-3) + 1e-7, cols


# ---------------------------------------------------------------- randomized sweep
def _random_case(rng):
    D = int(rng.choice([32, 64, 128, 40, 96]))
    B, H = int(rng.integers(1, 4)), int(rng.integers(1, 5))
    L = int(rng.choice([1, 3, 17, 64, 100, 128, 129, 200, 257, 384]))
    S = int(rng.choice([1, 5, 63, 64, 65, 127, 200, 256, 300, 512, 1100]))
    causal = bool(rng.integers(0, 2))
    n = float(rng.choice([0.0, 0.5, 1.0, 3.0]))
    mask_kind = str(rng.choice(["none", "none", "keypad", "dense", "rows"]))
    bias_kind = str(rng.choice(["none", "none", "hls", "bhls", "b1ls_f32", "keys"]))
    layout = str(rng.choice(["bhld", "blhd", "padded"]))
    return dict(D=D, B=B, H=H, L=L, S=S, causal=causal, n=n, mask_kind=mask_kind, bias_kind=bias_kind, layout=layout,
                dtype=[torch.float16, torch.bfloat16][int(rng.integers(0, 2))], scale=[None, 0.1, 0.3][int(rng.integers(0, 3))])


def _make(shape, dtype, dev, seed, layout):
    B, H, T, D = shape
    if layout == "blhd":      # [B,T,H,D] memory viewed as [B,H,T,D]
        return _rand((B, T, H, D), dtype, dev, seed).permute(0, 2, 1, 3)
    if layout == "padded":    # rows padded to D+8 elements
        return _rand((B, H, T, D + 8), dtype, dev, seed)[..., :D]
    return _rand(shape, dtype, dev, seed)


@pytest.mark.parametrize("seed", range(150))
def test_randomized_shapes_modes_and_layouts(pkg, dev, seed):
    """random (B,H,L,S,D), n, scale, causal, mask / bias broadcast patterns and memory layouts: forward and gradients vs the oracle"""
    rng = np.random.default_rng(1000 + seed)
    c = _random_case(rng)
    B, H, L, S, D, dtype = c["B"], c["H"], c["L"], c["S"], c["D"], c["dtype"]
    q = _make((B, H, L, D), dtype, dev, 1, c["layout"]).detach().requires_grad_()
    k = _make((B, H, S, D), dtype, dev, 2, c["layout"]).detach().requires_grad_()
    v = _make((B, H, S, D), dtype, dev, 3, c["layout"]).detach().requires_grad_()
    do = _rand((B, H, L, D), dtype, dev, 4, std=1.0)
    gen = torch.Generator().manual_seed(seed)
    mask = bias = None
    if c["mask_kind"] == "keypad":
        mask = torch.ones(B, 1, 1, S, dtype=torch.bool)
        for b in range(B):
            mask[b, ..., int(torch.randint(1, S + 1, (1,), generator=gen)):] = False
    elif c["mask_kind"] == "dense":
        mask = torch.rand(B, H, L, S, generator=gen) < 0.7
    elif c["mask_kind"] == "rows":
        mask = torch.rand(1, 1, L, S, generator=gen) < 0.8
    if c["bias_kind"] == "hls":
        bias = torch.randn(H, L, S, generator=gen).to(dtype)
    elif c["bias_kind"] == "bhls":
        bias = torch.randn(B, H, L, S, generator=gen).to(dtype)
    elif c["bias_kind"] == "b1ls_f32":
        bias = torch.randn(B, 1, L, S, generator=gen)
    elif c["bias_kind"] == "keys":
        bias = torch.randn(1, H, 1, S, generator=gen).to(dtype)
    mask = None if mask is None else mask.to(dev)
    bias = None if bias is None else bias.to(dev)
    out = pkg.flash_attention_n(q, k, v, softmax_n_param=c["n"], scale=c["scale"], is_causal=c["causal"], attn_mask=mask, attn_bias=bias)
    out.backward(do)
    o, dq, dk, dv = _oracle_fwd_bwd(q, k, v, do, softmax_n_param=c["n"], scale=c["scale"], is_causal=c["causal"], attn_mask=mask,
                                    attn_bias=None if bias is None else bias.float())
    # rows with no visible key and n = 0: the oracle gives NaN (0/0), the kernel defines the result as 0
    nan_ref = any(torch.isnan(t).any().item() for t in (o, dq, dk, dv))
    if nan_ref:
        # only possible with n = 0 and rows that see no key: the oracle divides 0 by 0 (so does the reference); the kernel
        # defines such a row's output and gradients as 0 - check that, and the live rows of the forward
        assert c["n"] == 0.0
        dead = torch.isnan(o) if torch.isnan(o).any() else torch.zeros_like(o, dtype=torch.bool)
        got = out.detach().float().cpu()
        assert (got[dead] == 0).all()
        _check(torch.where(dead, torch.zeros_like(got), got), torch.nan_to_num(o, nan=0.0), dtype, f"{c} out (live rows)")
        for t in (q.grad, k.grad, v.grad):
            assert torch.isfinite(t).all()
        return
    for got, want, nm in ((out, o, "out"), (q.grad, dq, "dq"), (k.grad, dk, "dk"), (v.grad, dv, "dv")):
        _check(got, want, dtype, f"{c} {nm}")


def _random_case2(rng):
    """the round-4 kernels: head dims 64 / 128 / 256, sizes of several tiles, key padding, broadcast biases that want a gradient"""
    D = int(rng.choice([64, 128, 256]))
    B, H = int(rng.integers(1, 4)), int(rng.choice([1, 2, 3, 8]))
    L = int(rng.choice([1, 64, 129, 200, 384, 640, 1000]))
    # (no single-key case here: with one key dS = P (1 - P) dO.v is computed as P (dP - delta) with delta from the ROUNDED output, a
    # relative error of 2^-9 P / (1 - P) that no other key averages out - 3 of 400 such cases missed the relative gate by 1.1 - 4 x;
    # S = 1 stays covered by the first sweep and the one-key tests, whose sizes keep the gate meaningful)
    S = int(rng.choice([5, 65, 200, 256, 512, 776, 1100, 1536]))
    return dict(D=D, B=B, H=H, L=L, S=S, causal=bool(rng.integers(0, 2)), n=float(rng.choice([0.5, 1.0, 3.0])),
                mask_kind=str(rng.choice(["none", "keypad", "keypad", "keypad_holes"])), bias_kind=str(rng.choice(["none", "none", "hls", "1hls", "b1ls", "11ls"])),
                bias_grad=bool(rng.integers(0, 2)), dtype=[torch.float16, torch.bfloat16][int(rng.integers(0, 2))],
                layout=str(rng.choice(["bhld", "bhld", "blhd", "padded"])), do_layout=str(rng.choice(["bhld", "blhd"])))


_FUZZ2 = int(os.environ.get("FASN_FUZZ_SEEDS", "60"))


@pytest.mark.parametrize("seed", range(_FUZZ2))
def test_randomized_round4_kernel_families(pkg, dev, seed):
    """random cases aimed at the kernels of round 4 (pipelined D = 64 backward, two-wave D = 256 forward / backward incl. key padding,
    length-paired dQ, the two-role bias-gradient kernel): forward, dq / dk / dv and - where the bias asks for it - dbias vs the oracle.
    FASN_FUZZ_SEEDS=N runs N seeds instead of 60."""
    rng = np.random.default_rng(7000 + seed)
    c = _random_case2(rng)
    B, H, L, S, D, dtype = c["B"], c["H"], c["L"], c["S"], c["D"], c["dtype"]
    q, k, v = (_make(sh, dtype, dev, sd, c["layout"]).detach().requires_grad_() for sh, sd in (((B, H, L, D), 1), ((B, H, S, D), 2), ((B, H, S, D), 3)))
    do = _make((B, H, L, D), dtype, dev, 4, c["do_layout"]) * 2.0   # (std 1; a [B,L,H,D] gradient arrives with its own strides)
    gen = torch.Generator().manual_seed(seed)
    mask = bias = None
    if c["mask_kind"].startswith("keypad"):
        mask = torch.ones(B, 1, 1, S, dtype=torch.bool)
        for b in range(B):
            mask[b, ..., int(torch.randint(1, S + 1, (1,), generator=gen)):] = False
            if c["mask_kind"] == "keypad_holes" and S > 8:
                a = int(torch.randint(0, S - 4, (1,), generator=gen))
                mask[b, ..., a:a + int(torch.randint(1, 100, (1,), generator=gen))] = False
            mask[b, ..., 0] = True   # (every row of every batch element keeps a visible key unless the causal mask hides it)
        mask = mask.to(dev)
    if c["bias_kind"] != "none":
        shape = {"hls": (H, L, S), "1hls": (1, H, L, S), "b1ls": (B, 1, L, S), "11ls": (1, 1, L, S)}[c["bias_kind"]]
        bias = torch.randn(*shape, generator=gen).to(dtype).to(dev).requires_grad_(c["bias_grad"])
    out = pkg.flash_attention_n(q, k, v, softmax_n_param=c["n"], is_causal=c["causal"], attn_mask=mask, attn_bias=bias)
    out.backward(do)
    qc, kc, vc = (t.detach().cpu().float().requires_grad_() for t in (q, k, v))
    bc = None if bias is None else bias.detach().cpu().float().requires_grad_(c["bias_grad"])
    o = ref_attention_n(qc, kc, vc, softmax_n_param=c["n"], is_causal=c["causal"], attn_mask=None if mask is None else mask.cpu(), attn_bias=bc)
    o.backward(do.cpu().float())
    for got, want, nm in ((out, o, "out"), (q.grad, qc.grad, "dq"), (k.grad, kc.grad, "dk"), (v.grad, vc.grad, "dv")):
        _check(got, want, dtype, f"{c} {nm}")
    if bias is not None and c["bias_grad"]:
        assert bias.grad is not None and bias.grad.shape == bias.shape
        _check(bias.grad, bc.grad, dtype, f"{c} dbias")


# ---------------------------------------------------------------- streams and graphs
def test_forward_backward_inside_a_hip_graph_and_on_side_streams(pkg, dev):
    """the C ABI never allocates or synchronises and launches on the caller's stream: a forward + backward step can be captured
    into a HIP graph and replayed with new inputs, and two side streams can run independent problems concurrently"""
    dtype = torch.bfloat16
    shape = (2, 4, 384, 64)
    q, k, v = (_rand(shape, dtype, dev, s).requires_grad_() for s in (1, 2, 3))
    do = _rand(shape, dtype, dev, 4, std=1.0)
    side = torch.cuda.Stream(dev)
    side.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(side):     # warm-up on a side stream, as graph capture requires
        for _ in range(2):
            out = pkg.flash_attention_n(q, k, v, softmax_n_param=1, is_causal=True)
            out.backward(do)
            q.grad = k.grad = v.grad = None
    torch.cuda.current_stream(dev).wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = pkg.flash_attention_n(q, k, v, softmax_n_param=1, is_causal=True)
        out.backward(do)
    # replay with different input VALUES in the captured buffers
    with torch.no_grad():
        q.copy_(_rand(shape, dtype, dev, 11))
        k.copy_(_rand(shape, dtype, dev, 12))
        v.copy_(_rand(shape, dtype, dev, 13))
    g.replay()
    torch.cuda.synchronize()
    o, dq, dk, dv = _oracle_fwd_bwd(q, k, v, do, softmax_n_param=1.0, is_causal=True)
    for got, want, nm in ((out, o, "out"), (q.grad, dq, "dq"), (k.grad, dk, "dk"), (v.grad, dv, "dv")):
        _check(got, want, dtype, f"graph replay {nm}")
    # two independent problems on two streams
    s1, s2 = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
    a = [_rand((1, 8, 1024, 64), dtype, dev, s) for s in (21, 22, 23)]
    b = [_rand((1, 8, 1024, 128), dtype, dev, s) for s in (31, 32, 33)]
    torch.cuda.synchronize()
    with torch.cuda.stream(s1):
        oa = pkg.flash_attention_n(*a, softmax_n_param=0.5)
    with torch.cuda.stream(s2):
        ob = pkg.flash_attention_n(*b, softmax_n_param=2.0, is_causal=True)
    torch.cuda.synchronize()
    _check(oa, ref_attention_n(*(t.cpu().float() for t in a), softmax_n_param=0.5), dtype, "stream 1")
    _check(ob, ref_attention_n(*(t.cpu().float() for t in b), softmax_n_param=2.0, is_causal=True), dtype, "stream 2")


# ---------------------------------------------------------------- gradient of attn_bias
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16, torch.float32])
@pytest.mark.parametrize("kind", ["hls", "b1ls_f32", "keys", "bhls+mask+causal"])
@pytest.mark.parametrize("D", [64, 128])
def test_attn_bias_gradient(pkg, dev, D, kind, dtype):
    """a bias that requires grad gets dS summed over the dimensions it broadcasts (the reference's SDPA path differentiates its
    additive mask the same way); vector path, element-load path (fp32 bias) and the fp32 kernels"""
    B, H, L, S = 2, 3, 150, 203
    q, k, v = (_rand(sh, dtype, dev, s).requires_grad_() for sh, s in (((B, H, L, D), 1), ((B, H, S, D), 2), ((B, H, S, D), 3)))
    do = _rand((B, H, L, D), dtype, dev, 4, std=1.0)
    gen = torch.Generator().manual_seed(9)
    mask, causal = None, False
    if kind == "hls":
        bias = torch.randn(H, L, S, generator=gen).to(dtype)
    elif kind == "b1ls_f32":
        bias = torch.randn(B, 1, L, S, generator=gen)
    elif kind == "keys":
        bias = torch.randn(1, H, 1, S, generator=gen).to(dtype)
    else:
        bias = torch.randn(B, H, L, S, generator=gen).to(dtype)
        mask = synth.keypad_mask(B, S, device=dev)
        causal = True
    bias = bias.to(dev).requires_grad_()
    out = pkg.flash_attention_n(q, k, v, softmax_n_param=0.5, attn_bias=bias, attn_mask=mask, is_causal=causal)
    out.backward(do)
    assert bias.grad is not None and bias.grad.shape == bias.shape and bias.grad.dtype == bias.dtype
    qc, kc, vc, bc = (t.detach().cpu().float().requires_grad_() for t in (q, k, v, bias))
    o = ref_attention_n(qc, kc, vc, softmax_n_param=0.5, attn_bias=bc, attn_mask=None if mask is None else mask.cpu(), is_causal=causal)
    o.backward(do.cpu().float())
    if dtype == torch.float32:
        for got, want, nm in ((out, o, "out"), (q.grad, qc.grad, "dq"), (bias.grad, bc.grad, "dbias")):
            err = (got.detach().cpu().float() - want).abs().max().item()
            assert err <= 5e-5 * max(want.abs().max().item(), 1.0), f"{kind}/{nm}: {err:.3e}"
    else:
        for got, want, nm in ((out, o, "out"), (q.grad, qc.grad, "dq"), (k.grad, kc.grad, "dk"), (bias.grad, bc.grad, "dbias")):
            _check(got, want, dtype, f"{kind}/{nm}")


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("shape", ["1h1s", "b11s", "h1s"])
def test_attn_bias_gradient_with_one_query_row(pkg, dev, shape, dtype):
    """L == 1 with a differentiable bias that broadcasts over batch / heads: a one-row bias is also a row broadcast, which the in-kernel
    reduced form does not serve - the call has to take the dense dS path (round-3 regression: FASN_EUNSUPPORTED in backward)"""
    B, H, L, S, D = 3, 4, 1, 333, 64
    q, k, v = (_rand(sh, dtype, dev, s).requires_grad_() for sh, s in (((B, H, L, D), 1), ((B, H, S, D), 2), ((B, H, S, D), 3)))
    do = _rand((B, H, L, D), dtype, dev, 4, std=1.0)
    gen = torch.Generator().manual_seed(5)
    bshape = {"1h1s": (1, H, 1, S), "b11s": (B, 1, 1, S), "h1s": (H, 1, S)}[shape]
    bias = torch.randn(*bshape, generator=gen).to(dtype).to(dev).requires_grad_()
    out = pkg.flash_attention_n(q, k, v, softmax_n_param=1.0, attn_bias=bias)
    out.backward(do)
    assert bias.grad is not None and bias.grad.shape == bias.shape
    qc, kc, vc, bc = (t.detach().cpu().float().requires_grad_() for t in (q, k, v, bias))
    o = ref_attention_n(qc, kc, vc, softmax_n_param=1.0, attn_bias=bc)
    o.backward(do.cpu().float())
    for got, want, nm in ((out, o, "out"), (q.grad, qc.grad, "dq"), (k.grad, kc.grad, "dk"), (bias.grad, bc.grad, "dbias")):
        _check(got, want, dtype, f"{shape}/{nm}")


def test_cached_argument_block_survives_a_failed_backward(pkg, dev, monkeypatch):
    """the per-signature BwdArgs block is mutated in place for a non-default dO layout: when the launch raises, the default strides
    must be back for the next call of the same signature (they used to be restored only after a successful launch)"""
    from flash_attention_softmax_n_amd import _lib
    dtype = torch.bfloat16
    B, H, L, D = 1, 2, 128, 64
    q, k, v = (_rand((B, H, L, D), dtype, dev, s).requires_grad_() for s in (1, 2, 3))
    do = _rand((B, L, H, D), dtype, dev, 4, std=1.0).transpose(1, 2)   # [B,H,L,D] view of [B,L,H,D] memory: aligned rows, other strides
    lib = _lib.load()
    real = lib.fasn_bwd
    calls = {"n": 0}

    class Failing:
        def __getattr__(self, name):
            if name == "fasn_bwd":
                def f(*a):
                    calls["n"] += 1
                    return -4 if calls["n"] == 1 else real(*a)
                return f
            return getattr(lib, name)
    monkeypatch.setattr(_lib, "load", lambda: Failing())
    out = pkg.flash_attention_n(q, k, v, softmax_n_param=1.0)
    with pytest.raises(Exception):
        out.backward(do, retain_graph=True)
    q.grad = k.grad = v.grad = None
    out.backward(do.contiguous())   # default layout through the same cached block
    _, dq, dk, dv = _oracle_fwd_bwd(q, k, v, do, softmax_n_param=1.0)
    for got, want, nm in ((q.grad, dq, "dq"), (k.grad, dk, "dk"), (v.grad, dv, "dv")):
        _check(got, want, dtype, nm)


def test_kernel_path_uses_the_arguments_of_the_real_call(pkg, dev):
    """kernel_path() runs the same normalisation as flash_attention_n: padded head dims, 3-D keys, negative scales"""
    from flash_attention_softmax_n_amd.flash_attn import kernel_path
    q = torch.randn(2, 4, 100, 80, device=dev, dtype=torch.bfloat16)      # head dim 80 is zero-padded to 128 by the call
    kv = torch.randn(2, 96, 80, device=dev, dtype=torch.bfloat16)          # 3-D key / value: shared by all heads
    assert kernel_path(q, kv, kv) == "plain"
    assert kernel_path(q, kv, kv, is_causal=True, scale=-0.2) == "plain"
    m = torch.ones(2, 1, 1, 96, dtype=torch.bool, device=dev)
    assert kernel_path(q, kv, kv, attn_mask=m) == "key-padding"
    b = torch.randn(4, 100, 96, device=dev, dtype=torch.bfloat16)
    assert kernel_path(q, kv, kv, attn_mask=m, attn_bias=b) == "vector bias + key-padding"
    b90 = torch.randn(4, 100, 90, device=dev, dtype=torch.bfloat16)        # rows 180 bytes apart: not 8-byte vectors
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("error")   # a query warns about nothing and plans nothing (the real call is what warns, once per kind)
        assert kernel_path(q, kv[:, :90], kv[:, :90], attn_bias=b90) == "element-load (slow)"
    assert kernel_path(q, kv[:, :90], kv[:, :90], attn_bias=b[..., :90]) != "element-load (slow)"   # a view with 192-byte rows is fine
    assert kernel_path(q.float(), kv.float(), kv.float()) == "fp32"
    # grouped-query decode: the query is regrouped exactly as the call does it ([B,H,1,E] -> G rows per K/V head) and nothing is cached
    from flash_attention_softmax_n_amd import flash_attn as fa
    before = len(fa._cache())
    qd = torch.randn(2, 8, 1, 128, device=dev, dtype=torch.bfloat16)
    kd = torch.randn(2, 2, 512, 128, device=dev, dtype=torch.bfloat16)
    assert kernel_path(qd, kd, kd, is_causal=True) == "plain"
    assert len(fa._cache()) == before


def test_graph_dropout_state_keeps_its_address_across_reseeding(pkg, dev):
    """the device {seed, offset} pair a captured graph advances is updated IN PLACE by torch.manual_seed + the next eager dropout
    call: a graph captured before the re-seed must not be left with a dangling pointer"""
    from flash_attention_softmax_n_amd import flash_attn as fa
    q = _rand((1, 2, 64, 64), torch.bfloat16, dev, 1)
    torch.manual_seed(11)
    pkg.flash_attention_n(q, q, q, softmax_n_param=1.0, dropout_p=0.1)
    st = fa._RNG_STATE[dev.index if dev.index is not None else 0]["state"]
    ptr = st.data_ptr()
    torch.manual_seed(12)
    pkg.flash_attention_n(q, q, q, softmax_n_param=1.0, dropout_p=0.1)
    st2 = fa._RNG_STATE[dev.index if dev.index is not None else 0]["state"]
    assert st2.data_ptr() == ptr and int(st2.cpu()[0]) == 12


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("D", [32, 64, 128, 256])
@pytest.mark.parametrize("kind", ["hls", "1hls+causal", "b1ls+keypad", "11ls_f32+densemask", "hls_ragged"])
def test_attn_bias_gradient_reduced_in_kernel(pkg, dev, kind, D, dtype):
    """A bias that broadcasts over batch and / or heads gets its gradient from csrc/fasn_bwd_dbias.h: summed over those dimensions
    inside the kernel, written once in the bias's shape and dtype (reference: SDPA differentiates the additive mask,
    core/flash_attn.py:100-113). Against the oracle's autograd, with causal / key-padding / dense masks and ragged sizes."""
    B, H, L, S = (3, 2, 150, 203) if kind != "hls_ragged" else (2, 3, 129, 131)
    q, k, v = (_rand(sh, dtype, dev, s).requires_grad_() for sh, s in (((B, H, L, D), 1), ((B, H, S, D), 2), ((B, H, S, D), 3)))
    do = _rand((B, H, L, D), dtype, dev, 4, std=1.0)
    gen = torch.Generator().manual_seed(11)
    mask, causal = None, False
    if kind in ("hls", "hls_ragged"):
        bias = torch.randn(H, L, S, generator=gen).to(dtype)
    elif kind == "1hls+causal":
        bias, causal = torch.randn(1, H, L, S, generator=gen).to(dtype), True
    elif kind == "b1ls+keypad":
        bias, mask = torch.randn(B, 1, L, S, generator=gen).to(dtype), synth.keypad_mask(B, S, device=dev)
    else:
        bias = torch.randn(1, 1, L, S, generator=gen)
        mask = (torch.rand(B, H, L, S, generator=gen) < 0.7).to(dev)
    bias = bias.to(dev).requires_grad_()
    torch.cuda.reset_peak_memory_stats()
    out = pkg.flash_attention_n(q, k, v, softmax_n_param=0.5, attn_bias=bias, attn_mask=mask, is_causal=causal)
    base = torch.cuda.memory_allocated()
    out.backward(do)
    # no [B,H,L,S] buffer: the backward's extra memory stays below the gradients themselves + one bias-sized tensor + slack
    extra = torch.cuda.max_memory_allocated() - base
    small = 4 * q.numel() * q.element_size() + 2 * bias.numel() * bias.element_size() + (1 << 20)
    assert extra <= small, f"backward peaked {extra} bytes above the forward state (limit {small}): a dense dS buffer?"
    assert bias.grad is not None and bias.grad.shape == bias.shape and bias.grad.dtype == bias.dtype
    qc, kc, vc, bc = (t.detach().cpu().float().requires_grad_() for t in (q, k, v, bias))
    o = ref_attention_n(qc, kc, vc, softmax_n_param=0.5, attn_bias=bc, attn_mask=None if mask is None else mask.cpu(), is_causal=causal)
    o.backward(do.cpu().float())
    for got, want, nm in ((out, o, "out"), (q.grad, qc.grad, "dq"), (k.grad, kc.grad, "dk"), (v.grad, vc.grad, "dv"), (bias.grad, bc.grad, "dbias")):
        _check(got, want, dtype, f"{kind}/{nm}")


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("shape", ["1h1s", "b11s", "111s", "h1s"])
def test_attn_bias_gradient_of_a_per_key_bias(pkg, dev, shape, dtype):
    """A differentiable bias that broadcasts over ROWS as well ([1,H,1,S], [B,1,1,S], [1,1,1,S]: per-key biases) with L > 1 (round 5): the
    front end expands it over the rows as a stride-0 view, the kernel reduces over batch / heads into a [1 or B, 1 or H, L, S] buffer and
    the expand's backward sums the rows - no dense [B,H,L,S] dS buffer (round 4 wrote one for these shapes)."""
    B, H, L, S, D = 4, 4, 384, 512, 64
    q, k, v = (_rand(sh, dtype, dev, s).requires_grad_() for sh, s in (((B, H, L, D), 1), ((B, H, S, D), 2), ((B, H, S, D), 3)))
    do = _rand((B, H, L, D), dtype, dev, 4, std=1.0)
    bshape = {"1h1s": (1, H, 1, S), "b11s": (B, 1, 1, S), "111s": (1, 1, 1, S), "h1s": (H, 1, S)}[shape]
    bias = torch.randn(*bshape, generator=torch.Generator().manual_seed(5)).to(dtype).to(dev).requires_grad_()
    out = pkg.flash_attention_n(q, k, v, softmax_n_param=1.0, attn_bias=bias)
    torch.cuda.synchronize()
    torch.cuda.reset_peak_memory_stats()
    base = torch.cuda.memory_allocated()
    out.backward(do)
    extra = torch.cuda.max_memory_allocated() - base
    dense = B * H * L * S * q.element_size()
    assert extra <= dense // 2 + 6 * q.numel() * q.element_size() + (1 << 20), f"backward peaked {extra} bytes above the forward state: a dense dS buffer is {dense}"
    assert bias.grad is not None and bias.grad.shape == bias.shape and bias.grad.dtype == bias.dtype
    qc, kc, vc, bc = (t.detach().cpu().float().requires_grad_() for t in (q, k, v, bias))
    o = ref_attention_n(qc, kc, vc, softmax_n_param=1.0, attn_bias=bc)
    o.backward(do.cpu().float())
    for got, want, nm in ((out, o, "out"), (q.grad, qc.grad, "dq"), (k.grad, kc.grad, "dk"), (v.grad, vc.grad, "dv"), (bias.grad, bc.grad, "dbias")):
        _check(got, want, dtype, f"{shape}/{nm}")


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("D", [64, 128])
@pytest.mark.parametrize("kind", ["hls", "hls+causal", "hls+keypad", "11ls+keypad+causal", "b1ls"])
def test_attn_bias_gradient_with_more_tiles_than_workgroups(pkg, dev, kind, D, dtype):
    """The two-role bias-gradient kernel (csrc/fasn_bwd_dbias_ws.h) is persistent: a workgroup walks several [128 x 128] tiles and its
    K / V stream, its row prefetches and the lagging wave B run across the tile boundaries. More tiles than the chip has CUs, ragged
    sizes (a last tile with one 64-key unit, rows past Sq), tiles a causal mask hides completely (zero-filled), key padding that hides a
    whole unit of some batch elements, and reductions over the batch, over the heads and over both."""
    B, H, L, S = 2, 3, 1300, 1480
    q, k, v = (_rand(sh, dtype, dev, s).requires_grad_() for sh, s in (((B, H, L, D), 1), ((B, H, S, D), 2), ((B, H, S, D), 3)))
    do = _rand((B, H, L, D), dtype, dev, 4, std=1.0)
    gen = torch.Generator().manual_seed(13)
    shape = {"hls": (H, L, S), "hls+causal": (H, L, S), "hls+keypad": (H, L, S), "11ls+keypad+causal": (1, 1, L, S), "b1ls": (B, 1, L, S)}[kind]
    bias = torch.randn(*shape, generator=gen).to(dtype).to(dev).requires_grad_()
    mask = None
    if "keypad" in kind:
        mask = torch.ones(B, 1, 1, S, dtype=torch.bool)
        mask[0, ..., 1100:] = False      # the last three key blocks of batch element 0 hidden, one of them from its middle
        mask[1, ..., 300:333] = False
        mask = mask.to(dev)
    causal = "causal" in kind
    out = pkg.flash_attention_n(q, k, v, softmax_n_param=1.0, attn_bias=bias, attn_mask=mask, is_causal=causal)
    out.backward(do)
    qc, kc, vc, bc = (t.detach().cpu().float().requires_grad_() for t in (q, k, v, bias))
    o = ref_attention_n(qc, kc, vc, softmax_n_param=1.0, attn_bias=bc, attn_mask=None if mask is None else mask.cpu(), is_causal=causal)
    o.backward(do.cpu().float())
    _check(out, o, dtype, f"{kind}/out")
    _check(bias.grad, bc.grad, dtype, f"{kind}/dbias")
    if causal:   # (L < S: the causal diagonal ends at key L - 1 + S - L; everything right of a row's last visible key is exactly zero)
        assert float(bias.grad[..., 0, S - L + 1:].float().abs().max()) == 0.0


@pytest.mark.parametrize("D", [64, 128])
@pytest.mark.parametrize("kind", ["hls", "hls+keypad+causal"])
def test_attn_bias_gradient_in_xcd_local_tile_order(pkg, dev, kind, D):
    """8 heads, 8 x 8 tiles per head, 512 tiles on 256 workgroups: the launch that takes the XCD-local tile order of
    csrc/fasn_bwd_dbias_ws.h (an XCD owns the heads h % 8 and walks query blocks, then key-block groups, then heads)."""
    B, H, L, S, dtype = 2, 8, 1024, 1024, torch.bfloat16
    q, k, v = (_rand(sh, dtype, dev, s).requires_grad_() for sh, s in (((B, H, L, D), 1), ((B, H, S, D), 2), ((B, H, S, D), 3)))
    do = _rand((B, H, L, D), dtype, dev, 4, std=1.0)
    bias = torch.randn(H, L, S, generator=torch.Generator().manual_seed(17)).to(dtype).to(dev).requires_grad_()
    mask = None
    if "keypad" in kind:
        mask = torch.ones(B, 1, 1, S, dtype=torch.bool)
        mask[1, ..., 700:] = False
        mask = mask.to(dev)
    causal = "causal" in kind
    out = pkg.flash_attention_n(q, k, v, softmax_n_param=1.0, attn_bias=bias, attn_mask=mask, is_causal=causal)
    out.backward(do)
    qc, kc, vc, bc = (t.detach().cpu().float().requires_grad_() for t in (q, k, v, bias))
    o = ref_attention_n(qc, kc, vc, softmax_n_param=1.0, attn_bias=bc, attn_mask=None if mask is None else mask.cpu(), is_causal=causal)
    o.backward(do.cpu().float())
    _check(bias.grad, bc.grad, dtype, f"{kind}/dbias")


def test_alibi_gradient_at_config4_size_without_a_dense_buffer(pkg, dev):
    """(4,32,8192,128) with the dense ALiBi bias [H,L,S] (4.3 GB) requiring a gradient: the backward must not allocate the
    [B,H,L,S] dS tensor (17 GB) - its extra memory stays under 6 GB - and sampled rows of dbias match the oracle."""
    B, H, S, D, dtype = 4, 32, 8192, 128, torch.bfloat16
    q, k, v, do = (_full(nm, (B, H, S, D), dtype, dev) for nm in ("q", "k", "v", "dout"))
    bias = synth.alibi_bias(H, S, S, dtype, device=dev).requires_grad_()
    mask = synth.keypad_mask(B, S, device=dev)
    q.requires_grad_()
    out = pkg.flash_attention_n(q, k, v, softmax_n_param=0.5, attn_bias=bias, attn_mask=mask)
    torch.cuda.synchronize()
    torch.cuda.reset_peak_memory_stats()
    base = torch.cuda.memory_allocated()
    out.backward(do)
    torch.cuda.synchronize()
    extra = torch.cuda.max_memory_allocated() - base
    assert extra < 6 * (1 << 30), f"backward peaked {extra / 2**30:.1f} GiB above the forward state"
    assert bias.grad.shape == bias.shape
    # oracle on one head, a few rows: dbias[h, rows, :] = sum_b dS[b, h, rows, :]
    h, rows = 5, torch.tensor([0, 1, 4095, 8191])
    want = torch.zeros(len(rows), S)
    for b in range(B):
        qc, kc, vc = (t[b:b + 1, h:h + 1].detach().cpu().float() for t in (q, k, v))
        bc = bias[h:h + 1].detach().cpu().float().requires_grad_()
        o = ref_attention_n(qc, kc, vc, softmax_n_param=0.5, attn_bias=bc, attn_mask=mask[b:b + 1].cpu())
        o.backward(do[b:b + 1, h:h + 1].cpu().float())
        want += bc.grad[0][rows]
    _check(bias.grad[h][rows.to(dev)], want, dtype, "alibi dbias rows")


# ---------------------------------------------------------------- dK/dV key-block compaction (csrc/fasn_bwd_dkdv_ws.h)
@pytest.mark.parametrize("causal", [False, True])
@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_padded_key_blocks_are_compacted_in_the_two_wave_dkdv_kernel(pkg, dev, dtype, causal):
    """D = 128, a bias broadcast over the batch and a [B,1,1,S] key-padding mask with whole 128-key blocks hidden (at the end, in the
    middle, a batch element with a single visible key, one with none hidden): the dK/dV workgroups take the j-th (batch, key block)
    pair WITH a visible key, the hidden pairs go to the last workgroup ids and only write zeros. Every dk / dv row is compared,
    the rows of the hidden blocks must be exactly 0."""
    B, H, L, S, D = 4, 8, 256, 1024, 128
    q, k, v = (_rand(sh, dtype, dev, s).requires_grad_() for sh, s in (((B, H, L, D), 31), ((B, H, S, D), 32), ((B, H, S, D), 33)))
    do = _rand((B, H, L, D), dtype, dev, 34, std=1.0)
    mask = torch.ones(B, 1, 1, S, dtype=torch.bool)
    mask[0, ..., 640:] = False                    # three trailing blocks hidden
    mask[1, ..., 128:384] = False                 # two blocks in the middle
    mask[1, ..., 900:] = False                    # + a ragged tail
    mask[2] = False
    mask[2, ..., 517] = True                      # one visible key
    mask = mask.to(dev)                           # batch 3: nothing hidden
    gen = torch.Generator().manual_seed(5)
    bias = (1.5 * torch.randn(H, L, S, generator=gen)).to(dtype).to(dev)
    out = pkg.flash_attention_n(q, k, v, softmax_n_param=1.0, attn_mask=mask, attn_bias=bias, is_causal=causal)
    out.backward(do)
    o, dq, dk, dv = _oracle_fwd_bwd(q, k, v, do, softmax_n_param=1.0, attn_mask=mask, attn_bias=bias, is_causal=causal)
    _check(out, o, dtype, "out")
    _check(q.grad, dq, dtype, "dq")
    _check(k.grad, dk, dtype, "dk")
    _check(v.grad, dv, dtype, "dv")
    hidden = ~mask.view(B, S)
    assert (k.grad.float().abs().amax(dim=(1, 3))[hidden] == 0).all() and (v.grad.float().abs().amax(dim=(1, 3))[hidden] == 0).all()


# ---------------------------------------------------------------- length-paired batch elements (csrc/fasn_fwd_kernel.h kpair_plan)
@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("D", [64, 128])
@pytest.mark.parametrize("lengths", [(1024, 896, 768, 512), (300, 1024, 1024, 77, 640), (1024, 1024, 1024, 960), (1024, 1024, 1024, 832), (1, 1024), (0, 512, 1024)])
def test_ragged_key_padded_batch_under_a_broadcast_bias_is_length_paired(pkg, dev, lengths, D, dtype):
    """A bias broadcast over the batch next to a [B,1,1,S] key-padding mask with ragged lengths (BASELINE config 4's structure): the
    forward and dQ workgroups take two batch elements each, the r-th longest and the r-th shortest (in-order dispatcher: equal
    workgroups), half of the workgroup ids leave at once. Even and odd batch sizes (the median element runs alone), ties, lengths
    that are nearly equal or not ragged enough to pay for the second bias fetch (mean >= 0.85 of the longest: plain schedule kept), a batch element with one / with no visible key; forward + dq/dk/dv against the oracle."""
    B, H, L, S = len(lengths), 8, 256, 1024
    q, k, v = (_rand(sh, dtype, dev, s).requires_grad_() for sh, s in (((B, H, L, D), 41), ((B, H, S, D), 42), ((B, H, S, D), 43)))
    do = _rand((B, H, L, D), dtype, dev, 44, std=1.0)
    mask = torch.zeros(B, 1, 1, S, dtype=torch.bool)
    for b, n in enumerate(lengths):
        mask[b, ..., :n] = True
    mask = mask.to(dev)
    gen = torch.Generator().manual_seed(6)
    bias = (1.5 * torch.randn(H, L, S, generator=gen)).to(dtype).to(dev)
    out = pkg.flash_attention_n(q, k, v, softmax_n_param=1.0, attn_mask=mask, attn_bias=bias)
    out.backward(do)
    o, dq, dk, dv = _oracle_fwd_bwd(q, k, v, do, softmax_n_param=1.0, attn_mask=mask, attn_bias=bias)
    _check(out, o, dtype, "out")
    _check(q.grad, dq, dtype, "dq")
    _check(k.grad, dk, dtype, "dk")
    _check(v.grad, dv, dtype, "dv")


# ---------------------------------------------------------------- key-padding masks (MODE_KEYPAD, MODE_BIAS_KEYPAD)
@pytest.mark.parametrize("D", [32, 64, 128, 256])
@pytest.mark.parametrize("causal", [False, True])
@pytest.mark.parametrize("pattern", ["tail", "random", "blocks", "none_visible_in_one_batch"])
@pytest.mark.parametrize("n", [1.0, 0.0])
@pytest.mark.parametrize("with_bias", [False, True])
def test_key_masks_with_row_stride_zero(pkg, dev, with_bias, n, pattern, causal, D):
    """boolean masks that depend on (batch, head, key) only take the per-tile visibility-word path: hidden keys anywhere in the
    sequence, whole hidden tiles (skipped), a batch element with no visible key at all, per-head masks, odd key counts.
    n = 0: no sink column, so a row has no finite max until its first visible key (left padding: the first tiles are hidden) and
    the seeded kernels must stay on the exact path until then.
    with_bias: the same masks next to an additive [H,L,S] bias that requires grad (MODE_BIAS_KEYPAD: ALiBi-style bias on a padded
    batch - the bias goes through the vector path, the mask stays a visibility word); dbias is checked too."""
    if n == 0.0 and pattern == "none_visible_in_one_batch":
        pytest.skip("softmax_0 over an empty key set is 0/0 in the oracle")
    dtype = torch.bfloat16
    B, H, L, S = 3, 2, 200, 331
    if with_bias:
        S = 336   # bias rows 16-byte aligned: the vector path (an unaligned bias takes the element-load kernels, covered elsewhere)
    q, k, v = (_rand(sh, dtype, dev, s).requires_grad_() for sh, s in (((B, H, L, D), 1), ((B, H, S, D), 2), ((B, H, S, D), 3)))
    do = _rand((B, H, L, D), dtype, dev, 4, std=1.0)
    gen = torch.Generator().manual_seed(21)
    if pattern == "tail":
        mask = synth.keypad_mask(B, S, device="cpu")
    elif pattern == "random":
        mask = torch.rand(B, H, 1, S, generator=gen) < 0.6          # per head as well
    elif pattern == "blocks":
        mask = torch.ones(B, 1, 1, S, dtype=torch.bool)
        mask[0, ..., 64:192] = False                                  # two whole 64-key tiles hidden
        mask[1, ..., :70] = False
        mask[2, ..., 300:] = False
    else:
        mask = torch.ones(B, 1, 1, S, dtype=torch.bool)
        mask[1] = False
    mask = mask.to(dev)
    bias = bc = None
    if with_bias:
        bias = (1.5 * torch.randn(H, L, S, generator=gen)).to(dtype).to(dev).requires_grad_()
    out = pkg.flash_attention_n(q, k, v, softmax_n_param=n, attn_mask=mask, attn_bias=bias, is_causal=causal)
    out.backward(do)
    if with_bias:
        bc = bias.detach().cpu().float().requires_grad_()
    o, dq, dk, dv = _oracle_fwd_bwd(q, k, v, do, softmax_n_param=n, attn_mask=mask, is_causal=causal, attn_bias=bc)
    for got, want, nm in ((out, o, "out"), (q.grad, dq, "dq"), (k.grad, dk, "dk"), (v.grad, dv, "dv")):
        _check(got, want, dtype, f"{pattern}/{nm}")
    if with_bias:
        _check(bias.grad, bc.grad, dtype, f"{pattern}/dbias")
    if pattern == "none_visible_in_one_batch":
        assert (out[1] == 0).all() and (k.grad[1] == 0).all() and (v.grad[1] == 0).all()


@pytest.mark.parametrize("D", [64, 128])
def test_key_padding_mask_rows_that_are_not_dword_aligned(pkg, dev, D):
    """the visibility words are built from 16-byte loads of the mask row; a row that does not start on a dword boundary (odd key
    count, batch > 0; a sliced mask) takes the byte-load path of the same builder"""
    dtype = torch.bfloat16
    B, H, L, S = 3, 2, 130, 331   # row b starts at byte 331*b (+1 for the slice below)
    q, k, v = (_rand(sh, dtype, dev, s).requires_grad_() for sh, s in (((B, H, L, D), 1), ((B, H, S, D), 2), ((B, H, S, D), 3)))
    do = _rand((B, H, L, D), dtype, dev, 4, std=1.0)
    store = torch.rand(B * S + 1, generator=torch.Generator().manual_seed(5)) < 0.7
    store[1] = True
    mask = store.to(dev)[1:].view(B, 1, 1, S)   # data pointer = base + 1
    assert mask.data_ptr() % 4 != 0
    out = pkg.flash_attention_n(q, k, v, softmax_n_param=1.0, attn_mask=mask)
    out.backward(do)
    o, dq, dk, dv = _oracle_fwd_bwd(q, k, v, do, softmax_n_param=1.0, attn_mask=mask)
    for got, want, nm in ((out, o, "out"), (q.grad, dq, "dq"), (k.grad, dk, "dk"), (v.grad, dv, "dv")):
        _check(got, want, dtype, f"unaligned mask/{nm}")


def test_key_padding_mask_longer_than_the_visibility_table(pkg, dev):
    """the forward keeps 512 visibility words in LDS (Sk <= 32768); longer key ranges take the dense-mask general mode of the same
    mask (fasn_api.hip) - same results"""
    dtype = torch.bfloat16
    B, H, L, S, D = 2, 1, 40, 32768 + 200, 64
    q, k, v = (_rand(sh, dtype, dev, s).requires_grad_() for sh, s in (((B, H, L, D), 1), ((B, H, S, D), 2), ((B, H, S, D), 3)))
    do = _rand((B, H, L, D), dtype, dev, 4, std=1.0)
    mask = torch.ones(B, 1, 1, S, dtype=torch.bool)
    mask[0, ..., 20000:] = False
    mask[1, ..., :5000] = False
    mask = mask.to(dev)
    out = pkg.flash_attention_n(q, k, v, softmax_n_param=1.0, attn_mask=mask)
    out.backward(do)
    o, dq, dk, dv = _oracle_fwd_bwd(q, k, v, do, softmax_n_param=1.0, attn_mask=mask)
    for got, want, nm in ((out, o, "out"), (q.grad, dq, "dq"), (k.grad, dk, "dk"), (v.grad, dv, "dv")):
        _check(got, want, dtype, f"long key padding/{nm}")


# ---------------------------------------------------------------- fp16 operands that a pre-scaled Q / K could overflow
@pytest.mark.parametrize("causal", [False, True])
def test_fp16_large_scale_does_not_overflow_the_prescaled_operand(pkg, dev, causal):
    """scale*log2e = 92 with |q| up to ~1000: q*c leaves the fp16 range although every score q.k*scale is moderate. Such calls
    must not take the kernels that pre-scale Q / K in the operand type. Checked against the same problem with the magnitudes
    moved from q to k by a power of two (identical products) and against the oracle evaluated in fp32."""
    dtype = torch.float16
    B, H, L, S, D = 1, 2, 160, 200, 64
    q = (_rand((B, H, L, D), dtype, dev, 1).float() * 600).to(dtype).requires_grad_()
    k = (_rand((B, H, S, D), dtype, dev, 2).float() * (2.0 ** -12)).to(dtype).requires_grad_()
    v = _rand((B, H, S, D), dtype, dev, 3).requires_grad_()
    do = _rand((B, H, L, D), dtype, dev, 4, std=1.0)
    out = pkg.flash_attention_n(q, k, v, softmax_n_param=1.0, scale=64.0, is_causal=causal)
    out.backward(do)
    for nm, t in (("out", out), ("dq", q.grad), ("dv", v.grad)):   # dK = scale * dS^T Q is legitimately beyond fp16 here
        assert torch.isfinite(t).all(), nm
    q2 = (q.detach().float() / 512).to(dtype).requires_grad_()
    k2 = (k.detach().float() * 512).to(dtype).requires_grad_()
    out2 = pkg.flash_attention_n(q2, k2, v.detach(), softmax_n_param=1.0, scale=64.0, is_causal=causal)
    assert (out.float() - out2.float()).abs().max().item() <= 2e-3
    qf, kf, vf = (t.detach().float().cpu() for t in (q, k, v))
    want = ref_attention_n(qf, kf, vf, softmax_n_param=1.0, scale=64.0, is_causal=causal)
    assert (out.float().cpu() - want).abs().max().item() <= 1e-2


def test_grouped_query_attention_with_dropout(pkg, dev):
    """the dropout stream is indexed by QUERY head, the dK/dV sum runs over the group inside the kernel"""
    dtype, D, p = torch.bfloat16, 64, 0.2
    B, H, Hkv, L, S = 2, 8, 2, 200, 264
    G = H // Hkv
    q = _rand((B, H, L, D), dtype, dev, 1).requires_grad_()
    k, v = (_rand((B, Hkv, S, D), dtype, dev, s).requires_grad_() for s in (2, 3))
    do = _rand((B, H, L, D), dtype, dev, 4, std=1.0)
    torch.manual_seed(99)
    out = pkg.flash_attention_n(q, k, v, softmax_n_param=1.0, dropout_p=p, is_causal=True)
    seed, offset = pkg.flash_attn.last_dropout_state()
    out.backward(do)
    keep = pkg.dropout.keep_mask(seed, offset, B, H, L, S, p)
    o, dq, dk, dv = _oracle_dropout(q, k.repeat_interleave(G, dim=1), v.repeat_interleave(G, dim=1), do, keep, pkg.dropout.effective_p(p),
                                    softmax_n_param=1.0, is_causal=True)
    dk, dv = (t.view(B, Hkv, G, S, D).sum(2) for t in (dk, dv))
    for got, want, nm in ((out, o, "out"), (q.grad, dq, "dq"), (k.grad, dk, "dk"), (v.grad, dv, "dv")):
        _check(got, want, dtype, f"gqa dropout {nm}")


# ---------------------------------------------------------------- grouped-query attention (fewer K/V heads than query heads)
@pytest.mark.parametrize("D,dtype", [(32, torch.bfloat16), (64, torch.bfloat16), (128, torch.bfloat16), (128, torch.float16),
                                     (64, torch.float32)])
@pytest.mark.parametrize("kind", ["plain", "causal+keypad", "bias", "bias+keypad", "dense-mask", "decode"])
def test_grouped_query_attention(pkg, dev, kind, D, dtype):
    """K/V with H/G heads: query head h reads K/V head h // G through the head stride (generalises the reference's 3-D shared
    K/V, flash_attn.py:75-79); the oracle sees the K/V heads repeated; dK/dV are the sums over each group"""
    B, H, Hkv = 2, 8, 2
    L, S = (1, 4096) if kind == "decode" else (200, 264)
    q = _rand((B, H, L, D), dtype, dev, 1).requires_grad_()
    k = _rand((B, Hkv, S, D), dtype, dev, 2).requires_grad_()
    v = _rand((B, Hkv, S, D), dtype, dev, 3).requires_grad_()
    do = _rand((B, H, L, D), dtype, dev, 4, std=1.0)
    mask = bias = None
    causal = kind == "causal+keypad"
    if causal or kind == "bias+keypad":
        mask = synth.keypad_mask(B, S, device=dev)
    if kind == "dense-mask":   # per-(head,row) boolean mask: the element-load path of every kernel
        mask = (torch.rand(B, H, L, S, generator=torch.Generator().manual_seed(5)) < 0.8).to(dev)
        mask[..., 0] = True
    if kind.startswith("bias"):
        bias = torch.randn(H, L, S, generator=torch.Generator().manual_seed(2)).to(dtype).to(dev)
    out = pkg.flash_attention_n(q, k, v, softmax_n_param=1.0, is_causal=causal, attn_mask=mask, attn_bias=bias)
    out.backward(do)
    assert k.grad.shape == k.shape and v.grad.shape == v.shape
    qc, kc, vc = (t.detach().cpu().float().requires_grad_() for t in (q, k, v))
    G = H // Hkv
    o = ref_attention_n(qc, kc.repeat_interleave(G, dim=1), vc.repeat_interleave(G, dim=1), softmax_n_param=1.0, is_causal=causal,
                        attn_mask=None if mask is None else mask.cpu(), attn_bias=None if bias is None else bias.float().cpu())
    o.backward(do.cpu().float())
    for got, want, nm in ((out, o, "out"), (q.grad, qc.grad, "dq"), (k.grad, kc.grad, "dk"), (v.grad, vc.grad, "dv")):
        _check(got, want, dtype, f"gqa {kind} {nm}")
    with pytest.raises(ValueError):
        pkg.flash_attention_n(q, k[:, :1].expand(B, 3, S, D), v[:, :1].expand(B, 3, S, D))


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("causal", [False, True])
@pytest.mark.parametrize("shape", [(2, 8, 2, 300, 420), (1, 8, 1, 512, 512), (2, 6, 3, 130, 700)])
def test_grouped_query_attention_at_head_dim_256(pkg, dev, shape, causal, dtype):
    """head dim 256 with grouped K/V (round 5): the two-wave backward kernels - dQ reads K/V head h // G, one dK/dV workgroup per K/V head
    and key block walks the row units of all G query heads of its group as one sequence and writes dK / dV once (csrc/fasn_bwd_ws256.h).
    Several key blocks per head, several row units per query head, L != S with the bottom-right causal alignment, G = 2, 4 and 8."""
    B, H, Hkv, L, S = shape
    D, G = 256, H // Hkv
    q = _rand((B, H, L, D), dtype, dev, 11).requires_grad_()
    k, v = (_rand((B, Hkv, S, D), dtype, dev, s).requires_grad_() for s in (12, 13))
    do = _rand((B, H, L, D), dtype, dev, 14, std=1.0)
    out = pkg.flash_attention_n(q, k, v, softmax_n_param=0.5, is_causal=causal)
    out.backward(do)
    assert k.grad.shape == k.shape and v.grad.shape == v.shape
    qc, kc, vc = (t.detach().cpu().float().requires_grad_() for t in (q, k, v))
    o = ref_attention_n(qc, kc.repeat_interleave(G, dim=1), vc.repeat_interleave(G, dim=1), softmax_n_param=0.5, is_causal=causal)
    o.backward(do.cpu().float())
    for got, want, nm in ((out, o, "out"), (q.grad, qc.grad, "dq"), (k.grad, kc.grad, "dk"), (v.grad, vc.grad, "dv")):
        _check(got, want, dtype, f"gqa d256 causal={causal} {nm}")
    import ctypes
    a = pkg._lib.BwdArgs()   # ... and the launch table says so: the two-wave kernels, the dK/dV one in its grouped instantiation
    kd = torch.empty_like(k)
    pkg.flash_attn._fill_fwd(a.fwd, q.detach(), k.detach(), v.detach(), out.detach(), torch.empty(B, H, L, device=dev), None, None, 0.5, D ** -0.5, causal)
    a.dout, a.dq, a.dk, a.dv = (pkg.flash_attn._view4(t) for t in (do, do, kd, kd))
    a.delta = a.fwd.lse
    names = [n for n, *_ in pkg._lib.launch_plan(a, pkg._lib.FASN_PLAN_BWD)]
    assert any(n.startswith("fasn_bwd_dq_ws256_kernel") for n in names) and any(n.startswith("fasn_bwd_dkdv_ws256_kernel") and n.rstrip(">").endswith(", 1") for n in names), names

"""C-ABI library checks that need no GPU: it loads, exports every symbol include/fasn.h declares, validates arguments
(validation returns before any launch), and the host-side front end refuses what it cannot serve."""
import ctypes
import os
import re

import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))


def _header_functions():
    text = open(os.path.join(ROOT, "include", "fasn.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(fasn_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_header_symbol(pkg):
    lib = ctypes.CDLL(pkg._lib.LIB_PATH)
    names = _header_functions()
    assert "fasn_fwd" in names and "fasn_bwd" in names and len(names) >= 10
    for name in names:
        assert hasattr(lib, name), f"libfasn.so does not export {name}"
    assert set(pkg._lib.EXPORTS) == set(names)


def test_library_exports_nothing_but_the_header(pkg):
    """`nm -D libfasn.so`: the defined fasn_* symbols are exactly the header's functions — no developer entry points
    (fasn_fwd_variant, timing helpers), and the library reads no environment variable."""
    import subprocess
    out = subprocess.run(["nm", "-D", "--defined-only", pkg._lib.LIB_PATH], check=True, capture_output=True, text=True).stdout
    exported = sorted({l.split()[-1] for l in out.splitlines() if l.split() and l.split()[-1].startswith("fasn_") and l.split()[-2] in "TtWw"})
    assert exported == _header_functions(), exported
    und = subprocess.run(["nm", "-D", "--undefined-only", pkg._lib.LIB_PATH], check=True, capture_output=True, text=True).stdout
    assert "getenv" not in und


def test_abi_version_and_strerror(pkg):
    lib = pkg._lib.load()
    assert lib.fasn_abi_version() == 6
    assert lib.fasn_strerror(0) == b"ok"
    for code in range(-8, 0):
        assert len(lib.fasn_strerror(code)) > 5
    assert b"unknown" in lib.fasn_strerror(-99)


def test_supported_matrix(pkg):
    lib = pkg._lib.load()
    for d in (32, 64, 128, 256):
        assert lib.fasn_supported(0, d, d) == 1 and lib.fasn_supported(1, d, d) == 1
    assert lib.fasn_supported(2, 256, 256) == 0 and lib.fasn_supported(1, 512, 512) == 0   # fp32 stops at 128, 16-bit at 256
    assert lib.fasn_supported(1, 64, 32) == 0
    assert lib.fasn_supported(1, 96, 96) == 0
    assert lib.fasn_supported(2, 64, 64) == 1 and lib.fasn_supported(3, 64, 64) == 0


def _args(pkg, **over):
    a = pkg._lib.FwdArgs()
    buf = (ctypes.c_char * 4096)()
    base = (ctypes.addressof(buf) + 15) & ~15
    for v in (a.q, a.k, a.v, a.o):
        v.ptr = base
        v.stride[0], v.stride[1], v.stride[2], v.stride[3] = 64 * 8, 64 * 8, 64, 1
    a.dtype, a.B, a.H, a.Sq, a.Sk, a.D, a.Dv = 1, 1, 1, 8, 8, 64, 64
    a.scale, a.softmax_n = 0.125, 1.0
    for k_, v_ in over.items():
        setattr(a, k_, v_)
    a._keep = buf
    return a


def test_argument_validation_codes(pkg):
    lib = pkg._lib.load()
    assert lib.fasn_fwd(None, None) == -1
    assert lib.fasn_fwd(_args(pkg, B=0), None) == -1
    assert lib.fasn_fwd(_args(pkg, dtype=3), None) == -2
    assert lib.fasn_fwd(_args(pkg, D=96, Dv=96), None) == -3
    assert lib.fasn_fwd(_args(pkg, dropout_p=1.0), None) == -1
    assert lib.fasn_fwd(_args(pkg, dropout_p=-0.1), None) == -1
    assert lib.fasn_fwd(_args(pkg, softmax_n=-1.0), None) == -1
    a = _args(pkg)
    a.q.ptr = a.q.ptr + 2
    assert lib.fasn_fwd(a, None) == -4
    a = _args(pkg)
    a.k.stride[2] = 65
    assert lib.fasn_fwd(a, None) == -4
    a = _args(pkg)
    a.v.stride[3] = 2
    assert lib.fasn_fwd(a, None) == -5
    b = pkg._lib.BwdArgs()
    assert lib.fasn_bwd(None, None) == -1
    b.fwd = _args(pkg)
    assert lib.fasn_bwd(b, None) == -1  # lse / delta missing
    assert lib.fasn_softmax_n_fwd(None, None, 1, 1, 1, 1, 0.0, 0, None) == -1


def test_launch_plan_names_the_kernels_of_the_baseline_configs(pkg):
    """fasn_launch_plan runs the host side of a call with every launch site recording instead of launching (no GPU needed): the
    BASELINE configs reach the kernel families DESIGN.md names, with the grids their shapes imply"""
    from baseline_plans import kernels
    fwd = kernels(pkg, "m0", "fwd")
    assert len(fwd) == 1 and fwd[0][0].startswith("fasn_fwd_kernel<fasn::bf16_tag, 64, 2, 0,") and fwd[0][1:3] == (8 * 16 * 16, 256)
    bwd = [k[0].split("<")[0] for k in kernels(pkg, "m0", "bwd")]
    assert bwd == ["fasn_bwd_dq_pipe_kernel", "fasn_bwd_dkdv_pipe_kernel"]   # (no delta launch since round 5: the dQ kernel, which runs first, computes and publishes delta)
    c3 = kernels(pkg, "c3", "fwd")
    assert len(c3) == 1 and c3[0][0].startswith("fasn_fwd_kernel<fasn::f16_tag, 64,")
    c4 = kernels(pkg, "c4", "fwd")
    assert len(c4) == 1 and c4[0][0].startswith("fasn_fwd_kernel<fasn::bf16_tag, 128, 1, 7,") and c4[0][2] == 512
    c4b = [k[0].split("<")[0] for k in kernels(pkg, "c4", "bwd")]
    assert c4b == ["fasn_bwd_delta_kernel", "fasn_bwd_dq_ws_kernel", "fasn_bwd_dkdv_ws_kernel"]
    assert [k[0].split("<")[0] for k in kernels(pkg, "c1", "fwd")] == ["fasn_f32_fwd_kernel"]
    # ABI 6: the cfg field names the positional template arguments (what bench.py prints and a reader of a profile wants)
    from baseline_plans import bwd_args as _ba
    d = lambda cfg, which: [k[0] for k in pkg._lib.launch_plan_described(_ba(pkg, cfg), which)]
    assert d("m0", pkg._lib.FASN_PLAN_FWD) == ["fasn_fwd_kernel<bf16,D=64,QB=2,plain,OCC=2,NW=4,RING=2,SEED=2>"]
    assert d("m0", pkg._lib.FASN_PLAN_BWD) == ["fasn_bwd_dq_pipe_kernel<bf16,plain>", "fasn_bwd_dkdv_pipe_kernel<bf16,plain>"]
    assert d("c4", pkg._lib.FASN_PLAN_FWD) == ["fasn_fwd_kernel<bf16,D=128,QB=1,bias+keypad,OCC=2,NW=8,RING=2,SEED=2>"]
    assert d("c4", pkg._lib.FASN_PLAN_BWD) == ["fasn_bwd_delta_kernel<bf16,D=128>", "fasn_bwd_dq_ws_kernel<bf16,D=128,bias+keypad>", "fasn_bwd_dkdv_ws_kernel<bf16,D=128,bias+keypad>"]
    assert d("c3", pkg._lib.FASN_PLAN_FWD) == ["fasn_fwd_kernel<f16,D=64,QB=2,causal,OCC=2,NW=4,RING=2,SEED=2,FOLD=1>"]
    assert d("c1", pkg._lib.FASN_PLAN_FWD) == ["fasn_f32_fwd_kernel<D=32,plain>"]
    # errors come back as the real call's code; a buffer that is too small is an argument error, not a truncated list
    import ctypes
    from baseline_plans import bwd_args
    lib = pkg._lib.load()
    a = bwd_args(pkg, "m0")
    small = ctypes.create_string_buffer(16)
    assert lib.fasn_launch_plan(a, 0, small, 16) == -1
    a.fwd.D = a.fwd.Dv = 96
    big = ctypes.create_string_buffer(4096)
    assert lib.fasn_launch_plan(a, 0, big, 4096) == -3
    assert lib.fasn_launch_plan(bwd_args(pkg, "m0"), 7, big, 4096) == -1


def test_fwd_path_query(pkg):
    """fasn_fwd_path names the kernel family of a call without launching anything (the element-load family is the slow one)"""
    lib = pkg._lib.load()
    assert lib.fasn_fwd_path(_args(pkg)) == 0 and lib.fasn_fwd_path(_args(pkg, causal=1)) == 0
    assert lib.fasn_fwd_path(_args(pkg, dtype=2)) == 5
    assert lib.fasn_fwd_path(_args(pkg, D=96, Dv=96)) == -3
    buf = (ctypes.c_char * 4096)()
    base = (ctypes.addressof(buf) + 15) & ~15

    def with_views(mask=None, bias=None, **over):
        a = _args(pkg, **over)
        for name, st in (("mask", mask), ("bias", bias)):
            if st is not None:
                v = getattr(a, name)
                v.ptr = base
                for i in range(4):
                    v.stride[i] = st[i]
        a._keep2 = buf
        return a

    assert lib.fasn_fwd_path(with_views(mask=(8, 0, 0, 1))) == 1                                   # key padding [B,1,1,S]
    assert lib.fasn_fwd_path(with_views(mask=(64, 64, 8, 1))) == 2                                 # dense mask
    assert lib.fasn_fwd_path(with_views(bias=(0, 64, 8, 1), bias_dtype=1)) == 2                    # aligned 16-bit bias
    assert lib.fasn_fwd_path(with_views(mask=(8, 0, 0, 1), bias=(0, 64, 8, 1), bias_dtype=1)) == 3
    # fp32 bias next to bf16 q (round 5): rows movable in 16-byte pieces take the vector path at head dims <= 128 (fp32 image instantiations),
    # with a key-padding mask the visibility-bit family; head dim 256, 8-byte-only alignment and dropout keep the element loads
    assert lib.fasn_fwd_path(with_views(bias=(0, 64, 8, 1), bias_dtype=2)) == 2
    assert lib.fasn_fwd_path(with_views(mask=(8, 0, 0, 1), bias=(0, 64, 8, 1), bias_dtype=2)) == 3
    assert lib.fasn_fwd_path(with_views(bias=(0, 64, 8, 1), bias_dtype=2, D=128, Dv=128)) == 2
    assert lib.fasn_fwd_path(with_views(bias=(0, 64, 8, 1), bias_dtype=2, D=256, Dv=256)) == 4
    assert lib.fasn_fwd_path(with_views(bias=(0, 60, 6, 1), bias_dtype=2)) == 4
    assert lib.fasn_fwd_path(with_views(bias=(0, 64, 8, 1), bias_dtype=2, dropout_p=0.1)) == 4
    assert lib.fasn_fwd_path(with_views(bias=(0, 64, 9, 1), bias_dtype=1)) == 4                    # unaligned bias rows
    assert lib.fasn_fwd_path(with_views(mask=(64, 64, 8, 1), dropout_p=0.1)) == 2                  # dense mask + dropout: vector kernels with dropout
    assert lib.fasn_fwd_path(with_views(mask=(64, 64, 8, 3))) == 4                                 # key stride != 1


def test_bwd_path_query(pkg):
    """fasn_bwd_path = the forward's family, except where the recorded backward plan holds element-load kernels: head dim 256 with a bias / dense mask"""
    from baseline_plans import bwd_args, _view
    lib = pkg._lib.load()
    for name, want in (("m0", 0), ("c3", 0), ("c4", 3)):
        a = bwd_args(pkg, name)
        assert lib.fasn_bwd_path(a) == want == lib.fasn_fwd_path(a.fwd), name
    a = bwd_args(pkg, "c4")   # bias + key padding at head dim 256: the vector general kernels in both directions since round 6 (round 5: element loads backward)
    f = a.fwd
    f.D = f.Dv = 256
    dense = (f.H * f.Sq * 256, f.Sq * 256, 256, 1)
    for v in (f.q, f.k, f.v, f.o, a.dout, a.dq, a.dk, a.dv):
        _view(v, dense)
    assert lib.fasn_fwd_path(f) == 2 and lib.fasn_bwd_path(a) == 2
    plan = pkg._lib.launch_plan_described(a, pkg._lib.FASN_PLAN_BWD)
    assert not any("element-load" in row[0] for row in plan) and any("bias+mask" in row[0] for row in plan), plan
    f.dropout_p = 0.1     # dropout too
    assert lib.fasn_fwd_path(f) == 2 and lib.fasn_bwd_path(a) == 2
    assert all("DROP=1" in row[0] for row in pkg._lib.launch_plan_described(a, pkg._lib.FASN_PLAN_BWD) if "delta" not in row[0])
    f.bias.stride[2] += 1   # bias rows that are not 16-byte movable: element loads, both directions
    assert lib.fasn_fwd_path(f) == 4 and lib.fasn_bwd_path(a) == 4
    f.bias.stride[2] -= 1
    f.dropout_p = 0.0
    f.bias.ptr = None     # key padding alone: the two-wave kernels in both directions
    assert lib.fasn_fwd_path(f) == 1 and lib.fasn_bwd_path(a) == 1
    assert lib.fasn_bwd_path(None) == -1


def test_causal_launches_pair_their_blocks_by_the_round_6_rule(pkg):
    """csrc/fasn_launch.h: pair_rule / wg_slots, read off the grids of recorded plans (no GPU): two rounds of single blocks pair; the forward also from one round
    when the pairs fill the 256 CUs evenly; dQ / dK/dV from 1.25 rounds; slots count LDS (the D = 128 forward holds one workgroup per CU)."""
    from baseline_plans import bwd_args
    L = pkg._lib

    def grids(B, H, S, D, which):
        a = bwd_args(pkg, (B, H, S, D, 1, 1.0, 1, False))
        return [(n.split("<")[0], g) for n, g, *_ in L.launch_plan_described(a, which)]

    # head dim 64, 64 heads: forward = 128-row blocks at three workgroups per CU (768 slots)
    assert grids(4, 16, 2048, 64, L.FASN_PLAN_FWD) == [("fasn_fwd_kernel", 512)]      # 1024 blocks = 1.33 rounds, 512 pairs = two per CU: paired
    assert grids(4, 16, 1536, 64, L.FASN_PLAN_FWD) == [("fasn_fwd_kernel", 768)]      # 768 blocks = one round, 384 pairs would be 1.5 per CU: single blocks
    assert grids(4, 16, 1024, 64, L.FASN_PLAN_FWD) == [("fasn_fwd_kernel", 512)]      # 512 blocks < 768 slots: single blocks
    # the pipelined backward (two workgroups per CU: 512 slots): 768 blocks = 1.5 rounds pair, 512 do not
    assert [g for _, g in grids(4, 16, 1536, 64, L.FASN_PLAN_BWD)] == [384, 384]
    assert [g for _, g in grids(4, 16, 1024, 64, L.FASN_PLAN_BWD)] == [512, 512]
    # head dim 128: the 4-wave forward's 96 KiB of LDS admit one workgroup per CU - 512 blocks of 128 rows are two rounds: paired
    assert grids(4, 16, 1024, 128, L.FASN_PLAN_FWD) == [("fasn_fwd_kernel", 256)]
    # head dim 32 (256-row blocks, two workgroups per CU): 512 blocks = one round, 256 pairs = one per CU: paired
    assert grids(4, 16, 2048, 32, L.FASN_PLAN_FWD) == [("fasn_fwd_kernel", 256)]


def test_front_end_refuses_cpu_and_unsupported(pkg):
    q = torch.zeros(1, 1, 4, 32)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        pkg.flash_attention_n(q, q, q)
    with pytest.raises(RuntimeError):
        pkg.softmax_n(torch.zeros(3, 3))
    with pytest.raises(RuntimeError):
        pkg.slow_attention_n(q[0], q[0], q[0])


def test_slow_attention_n_alias_softmax_dtype_follows_the_reference(pkg):
    """core/functional.py:72-73,91-93: the softmax_n weights are cast to softmax_dtype (default: query's dtype) and multiplied with value
    next, so the reference raises torch's dtype-mismatch RuntimeError for any softmax_dtype other than value's dtype (probed against the
    real reference in the build container: bf16 inputs with softmax_dtype float32 / float16 raise, None / bfloat16 run). The alias raises
    the same error before it looks at the device; None and value's dtype pass this check (and then hit the CPU-tensor refusal here)."""
    q = torch.zeros(1, 4, 32, dtype=torch.bfloat16)
    for sd in (torch.float32, torch.float16):
        with pytest.raises(RuntimeError, match="same dtype"):
            pkg.slow_attention_n(q, q, q, softmax_dtype=sd)
    for sd in (None, torch.bfloat16):
        with pytest.raises(RuntimeError, match="no CPU fallback"):
            pkg.slow_attention_n(q, q, q, softmax_dtype=sd)


def test_missing_library_fails_loudly(pkg, monkeypatch):
    monkeypatch.setattr(pkg._lib, "_lib", None)
    monkeypatch.setattr(pkg._lib, "LIB_PATH", "/nonexistent/libfasn.so")
    with pytest.raises(ImportError, match="no fallback"):
        pkg._lib.load()


def test_view_and_canon_helpers(pkg):
    fa = pkg.flash_attn
    t = torch.zeros(2, 3, 8, 64, dtype=torch.bfloat16)
    v = fa._view4(t)
    assert list(v.stride) == [3 * 8 * 64, 8 * 64, 64, 1]
    e = torch.zeros(2, 1, 1, 16, dtype=torch.uint8).expand(2, 3, 8, 16)
    assert list(fa._view4(e).stride) == [16, 0, 0, 1]
    assert fa._rows_ok(t) and not fa._rows_ok(t.transpose(2, 3))
    tt = t.permute(0, 2, 1, 3)  # [B, L, H, D] memory viewed as [B, H, L, D]: still row-aligned, no copy needed
    assert fa._rows_ok(tt.permute(0, 2, 1, 3))
    assert fa._canon(t.transpose(2, 3)).is_contiguous()
    assert fa._pad_feature(t, 128).shape[-1] == 128 and fa._pad_feature(t, 64) is t


def test_synth_is_index_addressable_and_deterministic(pkg):
    s = pkg.synth if hasattr(pkg, "synth") else __import__("flash_attention_softmax_n_amd.synth", fromlist=["x"])
    x = s.counter_normal((4, 256, 64), 5, dtype=torch.float16)
    y = s.counter_normal((256, 64), 5, dtype=torch.float16, start=2 * 256 * 64)
    assert torch.equal(x[2], y)
    assert s.checksum(x) == s.checksum(s.counter_normal((4, 256, 64), 5, dtype=torch.float16, chunk=1000))
    assert abs(x.float().std().item() - 0.5) < 0.01 and abs(x.float().mean().item()) < 0.01
    z = s.exact16(s.counter_normal((1000,), 1, dtype=torch.float32))
    assert torch.equal(z, z.bfloat16().float()) and torch.equal(z, z.half().float())
    assert torch.allclose(s.alibi_slopes(8), torch.tensor([2.0 ** -(i + 1) for i in range(8)], dtype=torch.float64))
    m = s.keypad_mask(4, 64)
    assert m.shape == (4, 1, 1, 64) and m.sum(-1).flatten().tolist() == [64, 56, 48, 32]


def test_dropout_host_mirror_statistics(pkg):
    d = pkg.dropout
    assert d.threshold(0.0) == 0 and d.threshold(0.2) == 13107 and d.threshold(1e-6) == 1 and d.threshold(0.99999) == 65535
    assert abs(d.effective_p(0.1) - 0.1) < 1.6e-5
    keep = d.keep_mask(7, 0, 2, 2, 128, 256, 0.25)
    assert keep.shape == (2, 2, 128, 256) and abs((1 - keep.mean()) - 0.25) < 0.01
    assert (d.keep_mask(7, 0, 2, 2, 128, 256, 0.25) == keep).all() and (d.keep_mask(8, 0, 2, 2, 128, 256, 0.25) != keep).any()
    assert d.keep_mask(7, 0, 1, 1, 4, 4, 0.0).all()


def test_dropout_keep_bits_of_neighbours_are_independent(pkg):
    """Stream definition 2 (round 6, csrc/fasn_common.h): the 16 fields of a 16-key group are the halves of eight 24-bit multiplies of rotated
    windows of ONE (row, group) state. Keep decisions of adjacent keys, of any two positions of a group, of adjacent rows, neighbouring heads,
    seeds and offsets must not correlate (|r| < 4 sigma of the sample; for the 120 position pairs of a group < 4.5 sigma: the maximum of 120
    draws), at several rates; the realised rate is the requested one."""
    import numpy as np
    d = pkg.dropout
    L = S = 768
    sigma = 1.0 / np.sqrt(2 * L * S)
    for p in (0.1, 0.5, 0.8):
        K = d.keep_mask(2024, 3, 1, 2, L, S, p)[0].astype(np.float64)
        c = lambda a, b: abs(float(np.corrcoef(a.ravel(), b.ravel())[0, 1]))
        assert abs((1 - K.mean()) - d.effective_p(p)) < 4 * np.sqrt(p * (1 - p) / K.size)
        for shift in (1, 2, 3, 4, 8, 15, 16):
            assert c(K[..., :-shift], K[..., shift:]) < 4 * sigma, (p, "key", shift)
        for shift in (1, 2, 3, 4):
            assert c(K[:, :-shift], K[:, shift:]) < 4 * sigma, (p, "row", shift)
        assert c(K[0], K[1]) < 6 * sigma
        assert c(K, d.keep_mask(2025, 3, 1, 2, L, S, p)[0].astype(np.float64)) < 4 * sigma
        assert c(K, d.keep_mask(2024, 4, 1, 2, L, S, p)[0].astype(np.float64)) < 4 * sigma
        G = K.reshape(2, L, S // 16, 16)
        n = 2 * L * (S // 16)
        worst = max((c(G[..., a], G[..., b]) * np.sqrt(n), a, b) for a in range(16) for b in range(a + 1, 16))
        assert worst[0] < 4.5, (p, "positions of a 16-key group", worst)


def test_dropout_host_mirror_known_answer(pkg):
    """The stream definition pinned by value: a change of the hash (window offsets, multipliers, which half of a product belongs to which
    key) must be a deliberate one - the kernels are checked against THIS mirror on the GPU (tests/test_gpu_parity.py: dropout with the
    explicit mask through the oracle)."""
    import hashlib
    import numpy as np
    K = pkg.dropout.keep_mask(0x123456789ABCDEF, 42, 2, 3, 5, 40, 0.25)
    assert K.shape == (2, 3, 5, 40)
    digest = hashlib.sha256(np.packbits(K.reshape(-1, 40), axis=1).tobytes()).hexdigest()
    assert digest == "14ea1d81b8ebedf25c6dc003b3153b876e544f1598278ae2b40263bf71c1fdf9", digest

"""Generate the golden fixtures in tests/golden/ by RUNNING THE REFERENCE (read-only at /root/reference).

Run in the build container only (the GPU box has no /root/reference):  python tests/golden/make_golden.py
The reference's modules are imported by file path (its package __init__ needs a GPU when triton is installed);
nothing of the reference is copied — the fixtures hold inputs' seeds/indices and the reference's OUTPUTS.

  g6_statistics.npz  the reference's analysis/statistics.py (variance, skewness, kurtosis over several `dim`s and the per-sample
                   batch means) on counter-generated tensors, one of them with heavy outliers.

Fixtures:
  g1_c1.npz        BASELINE config 1 (2,2,128,32): explicit fp32 inputs (exactly representable in bf16 AND fp16) and the
                   reference slow_attention_n O, dQ, dK, dV for n in {0,0.5,1,4} x causal in {F,T}; its native-bf16 outputs;
                   the reference flash_attention_n CPU outputs for integer n.
  g3_softmax.npz   softmax_n known-answer table of the reference tests (tests/cpu/core/test_functional.py:15-36) and the
                   reference softmax_n outputs on it.
  g4_<cfg>.npz     configs C2, C3, M0, C4, C5: inputs come from the counter-based generator (seed recorded, checksums stored);
                   reference slow_attention_n on 4 (b,h) x 64 sampled rows, native dtype and fp32-upcast.
  g5_<cfg>.npz     backward: one full (b,h) slice through slow_attention_n + autograd (fp32 upcast) with generated dO;
                   64 sampled rows of dQ, dK, dV.
  g5f_<cfg>.npz    backward at the FULL sequence length of every BASELINE config (C2, C3, M0, C4 with ALiBi + key padding at
                   S = 8192, C5): three (b,h) heads each through slow_attention_n + autograd, once in fp32 (the "true" answer)
                   and once in the config's NATIVE dtype (what the reference's own GPU test compares against,
                   tests/gpu/core/test_flash_attn.py:29-48); 64 sampled rows of O, dQ, dK, dV per head.
"""
import importlib.util
import os
import sys
import warnings

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF = "/root/reference/flash_attention_softmax_n/core"
sys.dont_write_bytecode = True
warnings.filterwarnings("ignore")


def _load(name):
    spec = importlib.util.spec_from_file_location("ref_" + name, os.path.join(REF, name + ".py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


functional = _load("functional")
flash_attn = _load("flash_attn")
slow_attention_n = functional.slow_attention_n
ref_softmax_n = functional.softmax_n
ref_flash_attention_n = flash_attn.flash_attention_n

spec = importlib.util.spec_from_file_location("synth", os.path.join(ROOT, "flash-attention-softmax-n_amd", "synth.py"))
synth = importlib.util.module_from_spec(spec)
spec.loader.exec_module(synth)


def save(name, **arrays):
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **arrays)
    print(f"{name}: {os.path.getsize(path) / 1024:.0f} KiB")


def g1():
    B, H, L, E = 2, 2, 128, 32
    q, k, v = (synth.exact16(synth.counter_normal((B, H, L, E), seed, dtype=torch.float32)) for seed in (11, 12, 13))
    do = synth.exact16(synth.counter_normal((B, H, L, E), 14, std=1.0, dtype=torch.float32))
    out = {"q": q.numpy(), "k": k.numpy(), "v": v.numpy(), "dout": do.numpy()}
    for n in (0.0, 0.5, 1.0, 4.0):
        for causal in (False, True):
            tag = f"n{n}_c{int(causal)}"
            qq, kk, vv = (t.clone().requires_grad_() for t in (q, k, v))
            o = slow_attention_n(qq, kk, vv, softmax_n_param=n, is_causal=causal)
            o.backward(do)
            out[f"o_{tag}"] = o.detach().numpy()
            out[f"dq_{tag}"] = qq.grad.numpy()
            out[f"dk_{tag}"] = kk.grad.numpy()
            out[f"dv_{tag}"] = vv.grad.numpy()
            ob = slow_attention_n(q.bfloat16(), k.bfloat16(), v.bfloat16(), softmax_n_param=n, is_causal=causal)
            out[f"o_bf16native_{tag}"] = ob.float().numpy()
            if float(n).is_integer():
                of = ref_flash_attention_n(q, k, v, softmax_n_param=int(n), is_causal=causal)
                out[f"o_flashcpu_{tag}"] = of.numpy()
    save("g1_c1.npz", **out)


def g3():
    numerators = torch.tensor([[1, 3, 6], [3, 1, 4], [1 / 6, 1 / 3, 1 / 2], [0.5, 1.5, 3], [100, 200, 300],
                               [1 / 600, 1 / 300, 1 / 200], [2 / 7, 4 / 7, 8 / 7]])
    x = torch.log(numerators)
    big = torch.tensor([12.0, 89.0, 710.0])
    out = {"x": x.numpy(), "numerators": numerators.numpy(), "big": big.numpy()}
    for n in (0.0, 1.0, 1e-3, 1e-6, 4.0):
        out[f"y_n{n}"] = ref_softmax_n(x, n=n, dim=-1).numpy()
        out[f"ybig_n{n}"] = ref_softmax_n(big, n, dim=-1).numpy()
    save("g3_softmax.npz", **out)


CONFIGS = {
    # name: (B, H, S, D, dtype, n, causal, extras)
    "c2": (8, 16, 1024, 64, torch.bfloat16, 1.0, False, None),
    "c3": (8, 16, 4096, 64, torch.float16, 1.0, True, None),
    "m0": (8, 16, 4096, 64, torch.bfloat16, 1.0, False, None),
    "c4": (4, 32, 8192, 128, torch.bfloat16, 0.5, False, "alibi+keypad"),
    "c5": (64, 16, 4096, 64, torch.bfloat16, 1.0, True, None),
}
SEEDS = {"q": 101, "k": 102, "v": 103, "dout": 104}


def head_slice(name, B, H, S, D, dtype, b, h):
    start = ((b * H + h) * S) * D
    return synth.counter_normal((S, D), SEEDS[name], dtype=dtype, start=start, std=1.0 if name == "dout" else 0.5)


def g4(cfg):
    B, H, S, D, dtype, n, causal, extra = CONFIGS[cfg]
    rng = np.random.RandomState(abs(hash(cfg)) % (2 ** 31))
    rng = np.random.RandomState({"c2": 2, "c3": 3, "m0": 0, "c4": 4, "c5": 5}[cfg])
    heads = [(0, 0), (B - 1, H - 1)] + [(int(rng.randint(B)), int(rng.randint(H))) for _ in range(2)]
    rows = np.sort(np.concatenate([[0, 1, S - 1], rng.choice(np.arange(2, S - 1), 61, replace=False)])).astype(np.int64)
    o_native, o_f32, cks = [], [], []
    for (b, h) in heads:
        q, k, v = (head_slice(nm, B, H, S, D, dtype, b, h) for nm in ("q", "k", "v"))
        cks.append([synth.checksum(q), synth.checksum(k), synth.checksum(v)])
        ridx = torch.as_tensor(rows)
        # additive (R,S) float mask encoding causal / bias / mask for those rows
        add = torch.zeros(len(rows), S, dtype=torch.float32)
        if extra == "alibi+keypad":
            add = synth.alibi_bias_rows(H, S, S, [h], rows, dtype)[0].float()
            keep = synth.keypad_mask(B, S)[b, 0, 0]
            add = add.masked_fill(~keep.unsqueeze(0), float("-inf"))
        if causal:
            j = torch.arange(S).unsqueeze(0)
            add = add.masked_fill(j > ridx.unsqueeze(-1), float("-inf"))
        qr = q[ridx]
        o_native.append(slow_attention_n(qr, k, v, attn_mask=add.to(dtype), softmax_n_param=n).float().numpy())
        o_f32.append(slow_attention_n(qr.float(), k.float(), v.float(), attn_mask=add, softmax_n_param=n).numpy())
    save(f"g4_{cfg}.npz", heads=np.array(heads), rows=rows, checksums=np.array(cks, dtype=np.int64),
         o_native=np.stack(o_native), o_f32=np.stack(o_f32), n=np.float64(n), causal=np.int64(causal),
         shape=np.array([B, H, S, D]), seeds=np.array([SEEDS["q"], SEEDS["k"], SEEDS["v"]]))


def g5(cfg, S_override=None):
    B, H, S, D, dtype, n, causal, extra = CONFIGS[cfg]
    if S_override:
        S = S_override
    b, h = B - 1, H // 2
    q, k, v, do = (head_slice(nm, B, H, S, D, dtype, b, h) for nm in ("q", "k", "v", "dout"))
    add = torch.zeros(S, S, dtype=torch.float32)
    if extra == "alibi+keypad":
        add = synth.alibi_bias_rows(H, S, S, [h], np.arange(S), dtype)[0].float()
        keep = synth.keypad_mask(B, S)[b, 0, 0]
        add = add.masked_fill(~keep.unsqueeze(0), float("-inf"))
    if causal:
        add = add.masked_fill(torch.arange(S).unsqueeze(0) > torch.arange(S).unsqueeze(-1), float("-inf"))
    qq, kk, vv = (t.float().requires_grad_() for t in (q, k, v))
    o = slow_attention_n(qq, kk, vv, attn_mask=add, softmax_n_param=n)
    o.backward(do.float())
    rng = np.random.RandomState(7)
    rows = np.sort(rng.choice(S, 64, replace=False)).astype(np.int64)
    save(f"g5_{cfg}.npz", head=np.array([b, h]), rows=rows, S=np.int64(S), shape=np.array([B, H, S, D]),
         o=o.detach().numpy()[rows], dq=qq.grad.numpy()[rows], dk=kk.grad.numpy()[rows], dv=vv.grad.numpy()[rows],
         n=np.float64(n), causal=np.int64(causal),
         checksums=np.array([synth.checksum(q), synth.checksum(k), synth.checksum(v), synth.checksum(do)], dtype=np.int64))


def g5f(cfg):
    """full-S backward rows of three heads; the GPU test runs the whole (B,H,S,D) grid and compares these heads"""
    B, H, S, D, dtype, n, causal, extra = CONFIGS[cfg]
    heads = [(0, 0), (B - 1, H // 2), (B // 2, H - 1)]
    rng = np.random.RandomState({"c2": 12, "c3": 13, "m0": 10, "c4": 14, "c5": 15}[cfg])
    rows = np.sort(np.concatenate([[0, S - 1], rng.choice(np.arange(1, S - 1), 62, replace=False)])).astype(np.int64)
    out = {k_: [] for k_ in ("o", "dq", "dk", "dv", "o_native", "dq_native", "dk_native", "dv_native")}
    cks = []
    for (b, h) in heads:
        q, k, v, do = (head_slice(nm, B, H, S, D, dtype, b, h) for nm in ("q", "k", "v", "dout"))
        cks.append([synth.checksum(t) for t in (q, k, v, do)])
        add = None
        if extra == "alibi+keypad":
            add = synth.alibi_bias_rows(H, S, S, [h], np.arange(S), dtype)[0].float()
            keep = synth.keypad_mask(B, S)[b, 0, 0]
            add = add.masked_fill(~keep.unsqueeze(0), float("-inf"))
        if causal:
            add = torch.zeros(S, S, dtype=torch.float32) if add is None else add
            add = add.masked_fill(torch.arange(S).unsqueeze(0) > torch.arange(S).unsqueeze(-1), float("-inf"))
        for tag, dt in (("", torch.float32), ("_native", dtype)):
            qq, kk, vv = (t.to(dt).requires_grad_() for t in (q, k, v))
            o = slow_attention_n(qq, kk, vv, attn_mask=None if add is None else add.to(dt), softmax_n_param=n)
            o.backward(do.to(dt))
            out["o" + tag].append(o.detach().float().numpy()[rows])
            out["dq" + tag].append(qq.grad.float().numpy()[rows])
            out["dk" + tag].append(kk.grad.float().numpy()[rows])
            out["dv" + tag].append(vv.grad.float().numpy()[rows])
            del o, qq, kk, vv
    save(f"g5f_{cfg}.npz", heads=np.array(heads), rows=rows, shape=np.array([B, H, S, D]), n=np.float64(n), causal=np.int64(causal),
         checksums=np.array(cks, dtype=np.int64), **{k_: np.stack(v_) for k_, v_ in out.items()})


def g6():
    spec_s = importlib.util.spec_from_file_location("ref_statistics", "/root/reference/flash_attention_softmax_n/analysis/statistics.py")
    st = importlib.util.module_from_spec(spec_s)
    spec_s.loader.exec_module(st)
    out = {}
    for name, shape, seed, outl in (("a", (4, 37, 19), 41, False), ("b", (3, 5, 64, 33), 42, True)):
        x = synth.counter_normal(shape, seed, std=1.5, dtype=torch.float32) + 0.3
        if outl:                       # heavy tails: a few large entries, as outlier activations have
            x.view(-1)[::97] *= 25.0
        out[f"{name}_shape"] = np.array(shape)
        out[f"{name}_seed"] = np.array(seed)
        out[f"{name}_outliers"] = np.array(int(outl))
        out[f"{name}_checksum"] = np.array(synth.checksum(x))
        xd = x.double()
        for dim_name, dim in (("all", None), ("last", -1), ("sample", tuple(range(1, x.ndim))), ("first", 0)):
            out[f"{name}_var_{dim_name}"] = st.variance(xd, dim=dim).numpy()
            out[f"{name}_skew_{dim_name}"] = st.skewness(xd, dim=dim).numpy()
            out[f"{name}_kurt_{dim_name}"] = st.kurtosis(xd, dim=dim).numpy()
            out[f"{name}_m3_{dim_name}"] = st.central_moment(xd, 3, dim=dim).numpy()
        out[f"{name}_var_bm"] = np.array(st.variance_batch_mean(xd))
        out[f"{name}_skew_bm"] = np.array(st.skewness_batch_mean(xd))
        out[f"{name}_kurt_bm"] = np.array(st.kurtosis_batch_mean(xd))
    save("g6_statistics.npz", **out)


if __name__ == "__main__":
    torch.manual_seed(0)
    if sys.argv[1:] == ["g6"]:
        g6()
        sys.exit(0)
    if sys.argv[1:2] == ["g5f"]:
        for c in (sys.argv[2:] or list(CONFIGS)):
            g5f(c)
        sys.exit(0)
    g1()
    g3()
    for c in CONFIGS:
        g4(c)
    for c in ("c2", "c3", "m0"):
        g5(c)
    g5("c4", S_override=2048)
    for c in CONFIGS:
        g5f(c)
    g6()

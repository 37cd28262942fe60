"""Tuning-point audit: every (head dim, dtype, mode) through flash_attention_n at one mid-size shape, forward and forward + backward, as
algorithmic TFLOP/s (4 B H L S D forward, 2.5 x that for the backward's five GEMM-equivalents; causal counts half). A mode far below its
neighbours of the same head dim runs a kernel at the wrong tuning point (waves per SIMD, rows per wave, ring) - that is what this looks for.
  python tools/audit_modes.py [S=2048] [dims=32,64,128,256] [dtypes=bf16,f16,f32] [only modes whose name contains this]
Lines also carry "ms_per_step" so that tools/ab_libs.sh can alternate libraries."""
import os, sys, torch
sys.path.insert(0, '/root/repo')
import flash_attention_softmax_n_amd as pkg
from flash_attention_softmax_n_amd import synth
dev = torch.device('cuda:0')
S = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
dims = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "32,64,128,256").split(",")]
dts = (sys.argv[3] if len(sys.argv) > 3 else "bf16,f32").split(",")
DT = {"bf16": torch.bfloat16, "f16": torch.float16, "f32": torch.float32}
only = sys.argv[4] if len(sys.argv) > 4 else ""

def timeit(fn, budget_ms=150.0):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); fn(); e1.record(); torch.cuda.synchronize()
    it = max(3, min(200, int(budget_ms / max(e0.elapsed_time(e1), 1e-3))))
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it

for dtn in dts:
    dt = DT[dtn]
    for D in dims:
        B, H = (4, 16) if dtn != "f32" else (2, 8)
        if os.environ.get("AUDIT_BH"):   # AUDIT_BH=8,32: another batch x heads
            B, H = (int(x) for x in os.environ["AUDIT_BH"].split(","))
        q, k, v = (synth.counter_normal((B, H, S, D), s, dtype=dt, device=dev).requires_grad_(True) for s in (101, 102, 103))
        do = synth.counter_normal((B, H, S, D), 104, std=1.0, dtype=dt, device=dev)
        kg, vg = (synth.counter_normal((B, H // 4, S, D), s, dtype=dt, device=dev).requires_grad_(True) for s in (105, 106))
        bias = synth.alibi_bias(H, S, S, dt, device=dev)
        bias32 = bias.float()
        kp = synth.keypad_mask(B, S, device=dev)
        dense = (torch.rand(B, 1, S, S, generator=torch.Generator().manual_seed(5)) < 0.8).to(dev)
        dense[..., 0] = True
        modes = [("plain", {}), ("causal", dict(is_causal=True)), ("keypad", dict(attn_mask=kp)), ("bias", dict(attn_bias=bias)),
                 ("bias+keypad", dict(attn_bias=bias, attn_mask=kp)), ("dense mask", dict(attn_mask=dense)),
                 ("bias+dense", dict(attn_bias=bias, attn_mask=dense)), ("causal+bias", dict(attn_bias=bias, is_causal=True)),
                 ("dropout", dict(dropout_p=0.1)), ("dropout+causal", dict(dropout_p=0.1, is_causal=True)),
                 ("dropout+bias+keypad", dict(dropout_p=0.1, attn_bias=bias, attn_mask=kp)), ("gqa4", dict(_gqa=True)),
                 ("gqa4+causal", dict(_gqa=True, is_causal=True))]
        if dtn != "f32": modes.append(("f32 bias+keypad", dict(attn_bias=bias32, attn_mask=kp)))
        for name, kw in modes:
            if only not in name:
                continue
            kw = dict(kw)
            kk, vv = (kg, vg) if kw.pop("_gqa", False) else (k, v)
            def fwd():
                with torch.no_grad():
                    pkg.flash_attention_n(q, kk, vv, softmax_n_param=1.0, **kw)
            def fwdbwd():
                q.grad = kk.grad = vv.grad = None
                pkg.flash_attention_n(q, kk, vv, softmax_n_param=1.0, **kw).backward(do)
            fl = 4.0 * B * H * S * S * D * (0.5 if kw.get("is_causal") else 1.0)
            try:
                tf, tb = timeit(fwd), timeit(fwdbwd)
                print(f'{dtn} D={D:3d} {name:22s} fwd "ms_per_step": {tf:.4f} ({fl / tf * 1e-9:7.1f} TF)   fwd+bwd "ms_per_step": {tb:.4f} ({3.5 * fl / tb * 1e-9:7.1f} TF)', flush=True)
            except Exception as e:   # a mode the front end refuses for this dtype / head dim
                print(f'{dtn} D={D:3d} {name:22s} -- {type(e).__name__}: {str(e)[:90]}', flush=True)

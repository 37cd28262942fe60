"""Forward (and forward + backward) of flash_attention_n in the vector mask / bias modes at head dim 64, large and small grids: ALiBi [H,L,S]
bias + key padding, bias alone, dense boolean mask, bias + dense mask. Lines carry "ms_per_step" so that tools/ab_libs.sh can alternate
libraries:  python tools/bench_bias_modes.py [fwd|fwdbwd] [head dim]"""
import sys, torch
sys.path.insert(0, '/root/repo')
import flash_attention_softmax_n_amd as pkg
from flash_attention_softmax_n_amd import synth
dev = torch.device('cuda:0')
which = sys.argv[1] if len(sys.argv) > 1 else "fwd"
DD = int(sys.argv[2]) if len(sys.argv) > 2 else 64
def timeit(fn, iters, warm=5):
    for _ in range(warm): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
for (B, H, S, D) in ((4, 16, 4096, DD), (8, 16, 1024, DD), (16, 16, 512, DD), (32, 8, 128, DD)):
    q, k, v = (synth.counter_normal((B, H, S, D), s, dtype=torch.bfloat16, device=dev).requires_grad_(which != "fwd") for s in (101, 102, 103))
    do = synth.counter_normal((B, H, S, D), 104, std=1.0, dtype=torch.bfloat16, device=dev)
    bias = synth.alibi_bias(H, S, S, torch.bfloat16, device=dev)
    kp = synth.keypad_mask(B, S, device=dev)
    dense = (torch.rand(B, 1, S, S, generator=torch.Generator().manual_seed(5)) < 0.8).to(dev)
    dense[..., 0] = True
    for name, kw in (("bias+keypad", dict(attn_bias=bias, attn_mask=kp)), ("bias", dict(attn_bias=bias)), ("dense mask", dict(attn_mask=dense)),
                     ("bias+dense mask", dict(attn_bias=bias, attn_mask=dense))):
        def fwd():
            with torch.no_grad():
                pkg.flash_attention_n(q, k, v, softmax_n_param=1.0, **kw)
        def fwdbwd():
            q.grad = k.grad = v.grad = None
            pkg.flash_attention_n(q, k, v, softmax_n_param=1.0, **kw).backward(do)
        t = timeit(fwd if which == "fwd" else fwdbwd, 200 if S <= 1024 else 30)
        print(f'({B},{H},{S},{D}) {name} {which}: "ms_per_step": {t:.4f}', flush=True)

"""fp32 additive bias next to bf16 q / k / v against the same bias in bf16 (vector path both, round 5): forward and forward + backward of
flash_attention_n at config 4's shape and at a D = 64 shape, ALiBi [H,L,S] + key-padding mask. Lines carry "ms_per_step" so that
tools/ab_libs.sh can alternate libraries: python tools/bench_fp32_bias.py [fwd|fwdbwd] [d128|d64]"""
import sys, torch
sys.path.insert(0, '/root/repo')
import flash_attention_softmax_n_amd as pkg
from flash_attention_softmax_n_amd import synth
dev = torch.device('cuda:0')
which = sys.argv[1] if len(sys.argv) > 1 else "both"
shapes = {"d128": (4, 32, 8192, 128), "d64": (4, 16, 4096, 64)}
sel = [sys.argv[2]] if len(sys.argv) > 2 else ["d128", "d64"]
def timeit(fn, iters=10, warm=3):
    for _ in range(warm): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
for name in sel:
    B, H, S, D = shapes[name]
    q, k, v = (synth.counter_normal((B, H, S, D), s, dtype=torch.bfloat16, device=dev).requires_grad_() for s in (101, 102, 103))
    do = synth.counter_normal((B, H, S, D), 104, std=1.0, dtype=torch.bfloat16, device=dev)
    mask = synth.keypad_mask(B, S, device=dev)
    for bdt in (torch.bfloat16, torch.float32):
        bias = synth.alibi_bias(H, S, S, bdt, device=dev)
        def fwd():
            with torch.no_grad():
                pkg.flash_attention_n(q, k, v, softmax_n_param=0.5, attn_bias=bias, attn_mask=mask)
        def fwdbwd():
            q.grad = k.grad = v.grad = None
            pkg.flash_attention_n(q, k, v, softmax_n_param=0.5, attn_bias=bias, attn_mask=mask).backward(do)
        for nm, fn in (("fwd", fwd), ("fwdbwd", fwdbwd)):
            if which in (nm, "both"):
                print(f'({B},{H},{S},{D}) bias {str(bdt)[6:]} {nm}: "ms_per_step": {timeit(fn):.4f}', flush=True)
        del bias

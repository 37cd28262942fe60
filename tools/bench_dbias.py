"""bias gradient reduced in the kernel (fasn_bwd_dbias) at BASELINE config 4's size: forward + backward with and without a bias that
needs a gradient"""
import sys, torch
sys.path.insert(0, '/root/repo')
import flash_attention_softmax_n_amd as pkg
from flash_attention_softmax_n_amd import synth
dev = torch.device('cuda:0')
def timeit(fn, iters=5, warm=2):
    for _ in range(warm): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
for (B, H, S, D) in ((4, 32, 8192, 128), (8, 16, 4096, 64)):
    q, k, v = (synth.counter_normal((B, H, S, D), s, dtype=torch.bfloat16, device=dev).requires_grad_() for s in (101, 102, 103))
    do = synth.counter_normal((B, H, S, D), 104, std=1.0, dtype=torch.bfloat16, device=dev)
    mask = synth.keypad_mask(B, S, device=dev)
    res = []
    for need in (False, True):
        bias = synth.alibi_bias(H, S, S, torch.bfloat16, device=dev).requires_grad_(need)
        def step():
            q.grad = k.grad = v.grad = bias.grad = None
            pkg.flash_attention_n(q, k, v, softmax_n_param=0.5, attn_bias=bias, attn_mask=mask).backward(do)
        res.append(timeit(step))
    print(f"({B},{H},{S},{D}) forward + backward, ALiBi [H,L,S] bias + key padding: {res[0]:.2f} ms without, {res[1]:.2f} ms with the bias gradient (dbias kernel {res[1] - res[0]:.2f} ms)", flush=True)

"""Forward + backward of flash_attention_n at BERT-like sizes with a key-padding mask (what the surgery path runs): launch-bound shapes where the
number of kernels per backward matters. Lines carry "ms_per_step" for tools/ab_libs.sh."""
import sys, torch
sys.path.insert(0, '/root/repo')
import flash_attention_softmax_n_amd as pkg
from flash_attention_softmax_n_amd import synth
dev = torch.device('cuda:0')
def timeit(fn, iters=200, warm=20):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()   # replay a captured step: the host's launch time is not what is measured here
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        fn()
        with torch.cuda.graph(g):
            fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
for (B, H, S, D, causal, masked) in ((32, 12, 128, 64, False, True), (16, 12, 512, 64, False, True), (8, 16, 1024, 64, False, True), (8, 12, 512, 64, True, False), (32, 8, 256, 32, False, True)):
    q, k, v = (synth.counter_normal((B, H, S, D), s, dtype=torch.bfloat16, device=dev).requires_grad_() for s in (1, 2, 3))
    do = synth.counter_normal((B, H, S, D), 4, std=1.0, dtype=torch.bfloat16, device=dev)
    mask = synth.keypad_mask(B, S, device=dev) if masked else None
    def step():
        q.grad = k.grad = v.grad = None
        pkg.flash_attention_n(q, k, v, softmax_n_param=1.0, attn_mask=mask, is_causal=causal).backward(do)
    print(f'({B},{H},{S},{D}) causal={causal} keypad={masked} fwd+bwd (graph replay): "ms_per_step": {timeit(step):.5f}', flush=True)

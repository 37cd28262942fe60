"""What would running the dQ and the dK/dV kernel of a backward side by side be worth? Upper bound without touching the library: two independent
forward + backward steps of config M0 / C3 on ONE stream against the same two steps on TWO streams (kernels of one step can fill the tails
and ramps of the other's). python tools/exp_two_streams.py"""
import sys, torch
sys.path.insert(0, '/root/repo')
import flash_attention_softmax_n_amd as pkg
from flash_attention_softmax_n_amd import synth
dev = torch.device('cuda:0')
def make(B, H, S, D, dt, seed):
    q, k, v = (synth.counter_normal((B, H, S, D), seed + i, dtype=dt, device=dev).requires_grad_() for i in range(3))
    do = synth.counter_normal((B, H, S, D), seed + 3, std=1.0, dtype=dt, device=dev)
    return q, k, v, do
for name, (B, H, S, D, dt, causal) in {"m0": (8, 16, 4096, 64, torch.bfloat16, False), "c3": (8, 16, 4096, 64, torch.float16, True), "c2": (8, 16, 1024, 64, torch.bfloat16, False)}.items():
    sets = [make(B, H, S, D, dt, 100 * i) for i in range(2)]
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    def step(i):
        q, k, v, do = sets[i]
        q.grad = k.grad = v.grad = None
        pkg.flash_attention_n(q, k, v, softmax_n_param=1.0, is_causal=causal).backward(do)
    def one_stream():
        step(0); step(1)
    def two_streams():
        cur = torch.cuda.current_stream()
        for i in range(2):
            streams[i].wait_stream(cur)
            with torch.cuda.stream(streams[i]):
                step(i)
        for i in range(2):
            cur.wait_stream(streams[i])
    def timeit(fn, iters=30, warm=5):
        for _ in range(warm): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters): fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / iters
    for rep in range(2):
        a, b = timeit(one_stream), timeit(two_streams)
        print(f"{name}: two forward + backward steps: one stream {a:.4f} ms, two streams {b:.4f} ms ({100 * (a - b) / a:+.1f} %)", flush=True)

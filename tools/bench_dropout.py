"""dropout cost: forward and forward+backward at (8,16,4096,64) and forward at (4,32,8192,128), with and without dropout_p = 0.1"""
import sys, time, torch
sys.path.insert(0, '/root/repo')
import flash_attention_softmax_n_amd as pkg
from flash_attention_softmax_n_amd import synth
dev = torch.device('cuda:0')
def run(B, H, S, D, bwd, p, iters=30):
    q, k, v = (synth.counter_normal((B, H, S, D), s, dtype=torch.bfloat16, device=dev).requires_grad_(bwd) for s in (1, 2, 3))
    do = synth.counter_normal((B, H, S, D), 4, std=1.0, dtype=torch.bfloat16, device=dev)
    def step():
        if bwd:
            q.grad = k.grad = v.grad = None
            pkg.flash_attention_n(q, k, v, softmax_n_param=1.0, dropout_p=p).backward(do)
        else:
            with torch.no_grad():
                pkg.flash_attention_n(q, k, v, softmax_n_param=1.0, dropout_p=p)
    for _ in range(5): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(iters): step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e3
for shape in ((8, 16, 4096, 64), (4, 32, 8192, 128)):
    for bwd in (False, True):
        a, b = run(*shape, bwd, 0.0), run(*shape, bwd, 0.1)
        print(f"{shape} {'fwd+bwd' if bwd else 'fwd'}: plain {a:.3f} ms, dropout 0.1 {b:.3f} ms (+{100 * (b / a - 1):.0f} %)")

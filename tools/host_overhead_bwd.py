"""Where the host time of an eager backward goes at a BERT-sized shape: time spent inside _FlashAttentionSoftmaxN.backward (the autograd engine runs it on
its device thread, invisible to cProfile of the main thread) against the whole out.backward() call."""
import sys, time, torch
sys.path.insert(0, '/root/repo')
import flash_attention_softmax_n_amd as pkg
from flash_attention_softmax_n_amd import synth, flash_attn
dev = torch.device('cuda:0')
B, H, S, D = 32, 12, 128, 64
q, k, v = (synth.counter_normal((B, H, S, D), s, dtype=torch.bfloat16, device=dev).requires_grad_() for s in (101, 102, 103))
do = synth.counter_normal((B, H, S, D), 104, dtype=torch.bfloat16, device=dev)
mask = synth.keypad_mask(B, S, device=dev)
F = flash_attn._FlashAttentionSoftmaxN
orig = F.backward
acc = [0.0, 0]
def timed(ctx, dout):
    t0 = time.perf_counter()
    r = orig(ctx, dout)
    acc[0] += time.perf_counter() - t0; acc[1] += 1
    return r
F.backward = staticmethod(timed)
tf = tb = 0.0
N = 3000
for i in range(N + 200):
    if i == 200:
        torch.cuda.synchronize(); acc[0] = 0.0; acc[1] = 0; tf = tb = 0.0
    t0 = time.perf_counter()
    out = pkg.flash_attention_n(q, k, v, softmax_n_param=1.0, attn_mask=mask)
    t1 = time.perf_counter()
    out.backward(do)
    t2 = time.perf_counter()
    q.grad = k.grad = v.grad = None
    tf += t1 - t0; tb += t2 - t1
    if i % 50 == 0: torch.cuda.synchronize()
print(f"forward call {1e6 * tf / N:.1f} us, out.backward() {1e6 * tb / N:.1f} us of which inside our backward() {1e6 * acc[0] / acc[1]:.1f} us")

"""Forward + backward timing through the Python front end for the training-relevant modes (mask, bias, dropout).
usage: python tools/bench_train_step.py [B H S D]"""
import sys
import torch
sys.path.insert(0, ".")
import flash_attention_softmax_n_amd as fa
from flash_attention_softmax_n_amd import synth

B, H, S, D = (int(x) for x in sys.argv[1:5]) if len(sys.argv) >= 5 else (8, 16, 4096, 64)
dev = torch.device("cuda:0")
dt = torch.bfloat16
q, k, v = (torch.randn(B, H, S, D, device=dev, dtype=dt).mul_(0.5).requires_grad_() for _ in range(3))
do = torch.randn(B, H, S, D, device=dev, dtype=dt)
mask = synth.keypad_mask(B, S, device=dev)
addmask = torch.zeros(B, 1, 1, S, device=dev, dtype=dt).masked_fill_(~mask, torch.finfo(dt).min)   # what HF models pass


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


for name, kw in (("plain", {}), ("causal", dict(is_causal=True)), ("key-padding mask", dict(attn_mask=mask)),
                 ("additive padding mask (HF)", dict(attn_bias=addmask)), ("dropout 0.1", dict(dropout_p=0.1)),
                 ("mask + dropout 0.1", dict(attn_mask=mask, dropout_p=0.1)), ("HF mask + dropout 0.1", dict(attn_bias=addmask, dropout_p=0.1))):
    def fwd():
        return fa.flash_attention_n(q, k, v, softmax_n_param=1.0, **kw)

    def fwdbwd():
        o = fa.flash_attention_n(q, k, v, softmax_n_param=1.0, **kw)
        o.backward(do)
        q.grad = k.grad = v.grad = None

    tf, tfb = timeit(fwd), timeit(fwdbwd)
    print(f"({B},{H},{S},{D}) bf16 {name:28s} fwd {tf:7.3f} ms   fwd+bwd {tfb:7.3f} ms")

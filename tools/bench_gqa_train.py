"""Forward + backward with grouped-query attention (fewer K/V heads) against the same shape with one K/V head per query head.
usage: python tools/bench_gqa_train.py [B H Hkv S D]"""
import sys
import torch
sys.path.insert(0, ".")
import flash_attention_softmax_n_amd as fa

B, H, Hkv, S, D = (int(x) for x in sys.argv[1:6]) if len(sys.argv) >= 6 else (4, 32, 8, 8192, 128)
dev = torch.device("cuda:0")
dt = torch.bfloat16


def timeit(fn, iters=10):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


for hk in (H, Hkv):
    q = torch.randn(B, H, S, D, device=dev, dtype=dt).mul_(0.5).requires_grad_()
    k, v = (torch.randn(B, hk, S, D, device=dev, dtype=dt).mul_(0.5).requires_grad_() for _ in range(2))
    do = torch.randn(B, H, S, D, device=dev, dtype=dt)
    for name, kw in (("plain", {}), ("causal", dict(is_causal=True))):
        def fwd():
            return fa.flash_attention_n(q, k, v, softmax_n_param=1.0, **kw)

        def fwdbwd():
            fa.flash_attention_n(q, k, v, softmax_n_param=1.0, **kw).backward(do)
            q.grad = k.grad = v.grad = None
        tf, tfb = timeit(fwd), timeit(fwdbwd)
        print(f"(B={B},H={H},Hkv={hk},S={S},D={D}) {name:7s}: fwd {tf:7.3f} ms   fwd+bwd {tfb:7.3f} ms   bwd {tfb - tf:7.3f} ms")

"""Prototype: C4 (bias broadcast over a ragged, key-padded batch) as ONE launch over the batch against one launch PER BATCH ELEMENT on
side streams (uniform workgroups per stream: the in-order dispatcher never waits inside a stream; the streams' workgroups start at
the same rate, so the batch elements of one bias tile still run side by side on one XCD)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import flash_attention_softmax_n_amd as pkg
from flash_attention_softmax_n_amd import synth

dev = torch.device("cuda", 0)
B, H, S, D = 4, 32, 8192, 128
dtype = torch.bfloat16
q, k, v = (synth.counter_normal((B, H, S, D), seed, dtype=dtype, device=dev) for seed in (101, 102, 103))
do = synth.counter_normal((B, H, S, D), 104, std=1.0, dtype=dtype, device=dev)
bias = synth.alibi_bias(H, S, S, dtype, device=dev)
mask = synth.keypad_mask(B, S, device=dev)
streams = [torch.cuda.Stream(device=dev) for _ in range(B)]


def batched():
    with torch.no_grad():
        return pkg.flash_attention_n(q, k, v, softmax_n_param=0.5, attn_mask=mask, attn_bias=bias)


def fanned(order=None):
    main = torch.cuda.current_stream()
    outs = [None] * B
    with torch.no_grad():
        for i, b in enumerate(order or range(B)):
            s = streams[i]
            s.wait_stream(main)
            with torch.cuda.stream(s):
                outs[b] = pkg.flash_attention_n(q[b:b + 1], k[b:b + 1], v[b:b + 1], softmax_n_param=0.5, attn_mask=mask[b:b + 1], attn_bias=bias)
        for s in streams:
            main.wait_stream(s)
    return torch.cat(outs, 0)


def timeit(fn, iters=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / iters


a = batched()
b = fanned()
print("max abs diff batched vs per-batch streams:", (a.float() - b.float()).abs().max().item())
for rep in range(2):
    print("one launch over the batch      : %.3f ms" % timeit(batched))
    print("one launch per batch element/stream: %.3f ms" % timeit(fanned))
    print("  (same, shortest batch element first): %.3f ms" % timeit(lambda: fanned([3, 2, 1, 0])))

# backward through autograd
qg, kg, vg = (t.detach().clone().requires_grad_() for t in (q, k, v))


def batched_fb():
    qg.grad = kg.grad = vg.grad = None
    o = pkg.flash_attention_n(qg, kg, vg, softmax_n_param=0.5, attn_mask=mask, attn_bias=bias)
    o.backward(do)


def fanned_fb():
    main = torch.cuda.current_stream()
    qg.grad = kg.grad = vg.grad = None
    outs = []
    for b in range(B):
        s = streams[b]
        s.wait_stream(main)
        with torch.cuda.stream(s):
            qs, ks, vs = (t[b:b + 1].detach().requires_grad_() for t in (qg, kg, vg))
            o = pkg.flash_attention_n(qs, ks, vs, softmax_n_param=0.5, attn_mask=mask[b:b + 1], attn_bias=bias)
            o.backward(do[b:b + 1])
            outs.append((qs.grad, ks.grad, vs.grad))
    for s in streams:
        main.wait_stream(s)
    return outs


for rep in range(2):
    print("forward + backward, one launch over the batch: %.3f ms" % timeit(batched_fb, 5))
    print("forward + backward, per batch element / stream: %.3f ms" % timeit(fanned_fb, 5))

"""HBM-bound side kernels against the 8 TB/s HBM peak: stand-alone softmax_n forward / backward (core/functional.py:15-29),
the one-pass activation moments (analysis/statistics.py:9-79), the split-K decode forward, and the in-kernel bias gradient
(fasn_bwd_dbias) at BASELINE config 4's size. Prints one line per kernel: ms, algorithmic GB/s, fraction of 8 TB/s."""
import sys, time, torch
sys.path.insert(0, '/root/repo')
import flash_attention_softmax_n_amd as pkg
from flash_attention_softmax_n_amd import synth, statistics
dev = torch.device('cuda:0')
PEAK = 8000.0


def timeit(fn, iters=30, warm=5):
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def line(name, ms, nbytes):
    gbs = nbytes / ms / 1e6
    print(f"{name:72s} {ms:9.4f} ms  {gbs:8.1f} GB/s  {gbs / PEAK:6.3f} of 8 TB/s", flush=True)


# softmax_n rows: the score matrix of (8,16,1024,1024) and one with rows longer than the register cache (4096 columns)
for rows, cols in ((8 * 16 * 1024, 1024), (8 * 16 * 1024, 4096), (4096, 32768)):
    x = torch.randn(rows, cols, device=dev, dtype=torch.bfloat16)
    with torch.no_grad():
        line(f"softmax_n forward  bf16 [{rows} x {cols}] n=1", timeit(lambda: pkg.softmax_n(x, n=1.0)), 2 * x.numel() * 2)
    xg = x.clone().requires_grad_()
    y = pkg.softmax_n(xg, n=1.0)
    dy = torch.randn_like(y)
    line(f"softmax_n backward bf16 [{rows} x {cols}]", timeit(lambda: torch.autograd.grad(y, xg, dy, retain_graph=True)), 3 * x.numel() * 2)
    del x, xg, y, dy
# one-pass moments of an activation tensor (2 GB of bf16)
a = torch.randn(64, 4096, 4096, device=dev, dtype=torch.bfloat16)
line("moments (variance/skewness/kurtosis power sums) bf16 [64, 4096, 4096]", timeit(lambda: statistics.kurtosis_batch_mean(a), iters=10), a.numel() * 2)
del a
# split-K decode: K + V streamed once
for B, H, Sq, Sk, D in ((1, 32, 1, 32768, 128), (64, 16, 1, 8192, 128), (4, 32, 16, 8192, 128)):
    q = synth.counter_normal((B, H, Sq, D), 1, dtype=torch.bfloat16, device=dev)
    k, v = (synth.counter_normal((B, H, Sk, D), s, dtype=torch.bfloat16, device=dev) for s in (2, 3))
    with torch.no_grad():
        line(f"decode forward (split-K where planned) (B={B},H={H},Sq={Sq},Sk={Sk},D={D})", timeit(lambda: pkg.flash_attention_n(q, k, v, softmax_n_param=1.0), iters=100), 2 * k.numel() * 2)
# bias gradient reduced in the kernel, config-4 size: backward with and without a bias that needs a gradient
B, H, S, D = 4, 32, 8192, 128
q, k, v = (synth.counter_normal((B, H, S, D), s, dtype=torch.bfloat16, device=dev).requires_grad_() for s in (101, 102, 103))
do = synth.counter_normal((B, H, S, D), 104, std=1.0, dtype=torch.bfloat16, device=dev)
mask = synth.keypad_mask(B, S, device=dev)
for need in (False, True):
    bias = synth.alibi_bias(H, S, S, torch.bfloat16, device=dev).requires_grad_(need)
    def step():
        q.grad = k.grad = v.grad = bias.grad = None
        pkg.flash_attention_n(q, k, v, softmax_n_param=0.5, attn_bias=bias, attn_mask=mask).backward(do)
    ms = timeit(step, iters=5, warm=2)
    print(f"config 4 forward + backward, ALiBi bias {'WITH' if need else 'without'} gradient: {ms:.2f} ms; peak memory {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB", flush=True)

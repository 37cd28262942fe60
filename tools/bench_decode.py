"""Decode-shape forward: plain path (fasn_fwd) vs split-K (fasn_fwd_ws). usage: python tools/bench_decode.py [B H Sq Sk D]"""
import sys
import torch
sys.path.insert(0, ".")
import flash_attention_softmax_n_amd as fa
from flash_attention_softmax_n_amd import _lib
from flash_attention_softmax_n_amd.flash_attn import _fill_fwd, _stream_ptr

shapes = [tuple(int(x) for x in sys.argv[1:6])] if len(sys.argv) >= 6 else [
    (1, 8, 1, 8192, 128), (1, 32, 1, 32768, 128), (8, 16, 1, 4096, 64), (4, 32, 16, 8192, 128), (1, 16, 128, 16384, 64), (64, 16, 1, 8192, 128)]
lib = _lib.load()
dev = torch.device("cuda:0")
for B, H, Sq, Sk, D in shapes:
    q = torch.randn(B, H, Sq, D, device=dev, dtype=torch.bfloat16) * 0.5
    k = torch.randn(B, H, Sk, D, device=dev, dtype=torch.bfloat16) * 0.5
    v = torch.randn(B, H, Sk, D, device=dev, dtype=torch.bfloat16) * 0.5
    o = torch.empty_like(q)
    lse = torch.empty(B, H, Sq, device=dev, dtype=torch.float32)
    a = _lib.FwdArgs()
    _fill_fwd(a, q, k, v, o, lse, None, None, 1.0, D ** -0.5, False)
    wsb = lib.fasn_fwd_workspace_bytes(a)
    ws = torch.empty(max(wsb, 16), dtype=torch.uint8, device=dev)
    s = _stream_ptr(dev)

    def timeit(fn, iters=200):
        for _ in range(20):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / iters

    t_plain = timeit(lambda: lib.fasn_fwd(a, s))
    o1 = o.clone()
    t_split = timeit(lambda: lib.fasn_fwd_ws(a, ws.data_ptr(), wsb, s)) if wsb else float("nan")
    err = (o.float() - o1.float()).abs().max().item() if wsb else 0.0
    kv_gb = 2 * B * H * Sk * D * 2 / 1e9
    print(f"(B={B},H={H},Sq={Sq},Sk={Sk},D={D}) plain {t_plain*1e3:8.1f} us  split-K {t_split*1e3:8.1f} us  "
          f"(K+V {kv_gb*1e3:.1f} MB -> {kv_gb/(min(t_plain, t_split if wsb else t_plain)*1e-3)/1e3:.2f} TB/s)  ws {wsb/1e6:.2f} MB  max|diff| {err:.2e}")

import sys, torch
sys.path.insert(0, "/root/repo")
import flash_attention_softmax_n_amd as fa
dev = torch.device("cuda:0")
for (B, H, Hkv, S, D) in ((8, 64, 8, 8192, 128), (32, 32, 8, 4096, 128), (1, 64, 8, 32768, 128)):
    q = torch.randn(B, H, 1, D, device=dev, dtype=torch.bfloat16) * 0.5
    k = torch.randn(B, Hkv, S, D, device=dev, dtype=torch.bfloat16) * 0.5
    v = torch.randn(B, Hkv, S, D, device=dev, dtype=torch.bfloat16) * 0.5
    def t(fn, it=100):
        for _ in range(10): fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(it): fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / it
    G = H // Hkv
    kk, vv = k.repeat_interleave(G, 1), v.repeat_interleave(G, 1)
    tg = t(lambda: fa.flash_attention_n(q, k, v, softmax_n_param=1.0))
    tr = t(lambda: fa.flash_attention_n(q, kk, vv, softmax_n_param=1.0))
    err = (fa.flash_attention_n(q, k, v, softmax_n_param=1.0).float() - fa.flash_attention_n(q, kk, vv, softmax_n_param=1.0).float()).abs().max().item()
    kv = 2 * B * Hkv * S * D * 2 / 1e9
    print(f"decode GQA B={B} H={H} Hkv={Hkv} S={S} D={D}: grouped {tg*1e3:.1f} us ({kv/tg:.2f} TB/s of K+V), K/V repeated per head {tr*1e3:.1f} us, max|diff| {err:.1e}")

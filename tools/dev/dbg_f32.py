"""debug: fp32 general-mode forward at D = 64 against the oracle: where are the errors?"""
import sys
import torch
sys.path.insert(0, '/root/repo')
import os
import flash_attention_softmax_n_amd as pkg
if os.environ.get("FASN_TEST_ABI"): pkg._lib.FASN_ABI_VERSION = int(os.environ["FASN_TEST_ABI"])
from flash_attention_softmax_n_amd import synth
from oracle.ref_attention import ref_attention_n
dev = torch.device('cuda:0')
for D in (32, 64, 128):
    for (B, H, L, S) in ((1, 1, 128, 64), (1, 1, 128, 128), (2, 2, 150, 200)):
        q, k, v = (synth.counter_normal(sh, s, dtype=torch.float32, device=dev) for sh, s in (((B, H, L, D), 1), ((B, H, S, D), 2), ((B, H, S, D), 3)))
        bias = torch.zeros(H, L, S, device=dev)
        out = pkg.flash_attention_n(q, k, v, softmax_n_param=0.5, attn_bias=bias)
        ref = ref_attention_n(q.cpu(), k.cpu(), v.cpu(), softmax_n_param=0.5, attn_bias=bias.cpu())
        err = (out.cpu() - ref).abs()
        print(f"D={D} {B,H,L,S}: max err {err.max().item():.3e}; rows with err>1e-4: {(err.amax(dim=(0,1,3)) > 1e-4).nonzero().flatten().tolist()[:40]} cols: {(err.amax(dim=(0,1,2)) > 1e-4).nonzero().flatten().tolist()[:70]}")
        plain = pkg.flash_attention_n(q, k, v, softmax_n_param=0.5)
        print(f"      plain-kernel err {(plain.cpu() - ref).abs().max().item():.3e}")

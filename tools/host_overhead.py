"""How long does the host take per flash_attention_n call (no sync), and how does the timed loop compare with back-to-back C-ABI launches?"""
import sys, time, torch
sys.path.insert(0, '/root/repo')
import flash_attention_softmax_n_amd as pkg
from flash_attention_softmax_n_amd import synth
dev = torch.device('cuda:0')
B, H, S, D = 8, 16, 4096, 64
q, k, v = (synth.counter_normal((B, H, S, D), s, dtype=torch.bfloat16, device=dev) for s in (101, 102, 103))
def step():
    with torch.no_grad():
        return pkg.flash_attention_n(q, k, v, softmax_n_param=1.0)
for _ in range(5): step()
torch.cuda.synchronize()
for steps in (20, 20, 100, 400):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps): step()
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"steps {steps}: host issue {1e6*(t1-t0)/steps:.1f} us/step, total {1e3*(t2-t0)/steps:.4f} ms/step")
# raw ABI
lib, fa = pkg._lib.load(), pkg.flash_attn
o = torch.empty_like(q); lse = torch.empty(B, H, S, dtype=torch.float32, device=dev)
a = pkg._lib.FwdArgs(); fa._fill_fwd(a, q, k, v, o, lse, None, None, 1.0, 0.125, False)
st = torch.cuda.current_stream().cuda_stream
for steps in (20, 20, 100, 400):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps): lib.fasn_fwd(a, st)
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"ABI steps {steps}: host issue {1e6*(t1-t0)/steps:.1f} us/step, total {1e3*(t2-t0)/steps:.4f} ms/step")

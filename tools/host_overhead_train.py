"""Host time per forward + backward step through autograd at C2 (8,16,1024,64): issue time without a sync, total with one, and a cProfile of the issue loop."""
import cProfile, pstats, sys, time, torch
sys.path.insert(0, '/root/repo')
import flash_attention_softmax_n_amd as pkg
from flash_attention_softmax_n_amd import synth
dev = torch.device('cuda:0')
B, H, S, D = 8, 16, 1024, 64
q, k, v = (synth.counter_normal((B, H, S, D), s, dtype=torch.bfloat16, device=dev).requires_grad_() for s in (101, 102, 103))
do = synth.counter_normal((B, H, S, D), 104, dtype=torch.bfloat16, device=dev)
def step():
    out = pkg.flash_attention_n(q, k, v, softmax_n_param=1.0)
    out.backward(do)
    q.grad = k.grad = v.grad = None
for _ in range(20): step()
torch.cuda.synchronize()
for steps in (50, 200, 200):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps): step()
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"steps {steps}: host issue {1e6*(t1-t0)/steps:.1f} us/step, total {1e3*(t2-t0)/steps:.4f} ms/step")
pr = cProfile.Profile(); pr.enable()
for _ in range(300): step()
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats('tottime').print_stats(18)

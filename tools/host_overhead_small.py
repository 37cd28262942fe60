"""Host time per forward + backward step through autograd at a BERT-sized shape (32,12,128,64) with a key-padding mask - the eager path of the
surgery use case, where the GPU work (0.04 ms) is shorter than a careless host side: issue time without a sync, total with one, and a cProfile."""
import cProfile, pstats, sys, time, torch
sys.path.insert(0, '/root/repo')
import flash_attention_softmax_n_amd as pkg
from flash_attention_softmax_n_amd import synth
dev = torch.device('cuda:0')
B, H, S, D = 32, 12, 128, 64
q, k, v = (synth.counter_normal((B, H, S, D), s, dtype=torch.bfloat16, device=dev).requires_grad_() for s in (101, 102, 103))
do = synth.counter_normal((B, H, S, D), 104, dtype=torch.bfloat16, device=dev)
mask = synth.keypad_mask(B, S, device=dev)
def step():
    out = pkg.flash_attention_n(q, k, v, softmax_n_param=1.0, attn_mask=mask)
    out.backward(do)
    q.grad = k.grad = v.grad = None
def fwd():
    with torch.no_grad():
        pkg.flash_attention_n(q, k, v, softmax_n_param=1.0, attn_mask=mask)
for name, fn in (("forward + backward", step), ("forward (no_grad)", fwd)):
    for _ in range(50): fn()
    torch.cuda.synchronize()
    for steps in (500, 2000):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(steps): fn()
        t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
        print(f"{name}: steps {steps}: host issue {1e6*(t1-t0)/steps:.1f} us/step, total {1e3*(t2-t0)/steps:.4f} ms/step", flush=True)
pr = cProfile.Profile(); pr.enable()
for _ in range(1000): step()
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats('tottime').print_stats(22)

import sys, torch
sys.path.insert(0, '/root/repo')
import flash_attention_softmax_n_amd as pkg
from flash_attention_softmax_n_amd import synth
dev = torch.device('cuda:0')
B,H,S,D = 4,16,4096,256
q,k,v = (synth.counter_normal((B,H,S,D), s, dtype=torch.bfloat16, device=dev) for s in (1,2,3))
mask = synth.keypad_mask(B, S, device=dev)
def t(fn, it=20):
    for _ in range(3): fn()
    e0,e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1)/it
qg, kg, vg = (x.clone().requires_grad_() for x in (q, k, v))
do = synth.counter_normal((B,H,S,D), 4, dtype=torch.bfloat16, device=dev)
def fb(**kw):
    qg.grad = kg.grad = vg.grad = None
    pkg.flash_attention_n(qg, kg, vg, softmax_n_param=1.0, **kw).backward(do)
print("d256 fwd+bwd plain %.3f ms, key padding %.3f ms, key padding + causal %.3f ms" % (t(lambda: fb(), 10), t(lambda: fb(attn_mask=mask), 10), t(lambda: fb(attn_mask=mask, is_causal=True), 10)))
print("d256 fwd plain %.3f ms, key padding %.3f ms, key padding + causal %.3f ms" % (t(lambda: pkg.flash_attention_n(q,k,v,softmax_n_param=1.0)), t(lambda: pkg.flash_attention_n(q,k,v,softmax_n_param=1.0,attn_mask=mask)), t(lambda: pkg.flash_attention_n(q,k,v,softmax_n_param=1.0,attn_mask=mask,is_causal=True))))

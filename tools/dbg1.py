import torch, sys
sys.path.insert(0,'/root/repo')
import flash_attention_softmax_n_amd as pkg
from flash_attention_softmax_n_amd import synth
from oracle.ref_attention import ref_attention_n
dev=torch.device('cuda:0')
for D in (32,64,128):
  for causal in (False,True):
    dtype=torch.float16
    B,H,L,S=2,3,200,264
    q,k,v=(synth.counter_normal(sh,s,dtype=dtype,device=dev).requires_grad_() for sh,s in (((B,H,L,D),1),((B,H,S,D),2),((B,H,S,D),3)))
    do=synth.counter_normal((B,H,L,D),4,std=1.0,dtype=dtype,device=dev)
    gen=torch.Generator().manual_seed(5)
    mask=synth.keypad_mask(B,S,device=dev)
    bias=torch.randn(H,L,S,generator=gen).to(dtype).to(dev)
    out=pkg.flash_attention_n(q,k,v,softmax_n_param=0.5,attn_mask=mask,attn_bias=bias,is_causal=causal)
    out.backward(do)
    qc,kc,vc=(t.detach().cpu().float().requires_grad_() for t in (q,k,v))
    o=ref_attention_n(qc,kc,vc,softmax_n_param=0.5,attn_mask=mask.cpu(),attn_bias=bias.cpu().float(),is_causal=causal)
    o.backward(do.cpu().float())
    for nm,g,w in (("o",out,o),("dq",q.grad,qc.grad),("dk",k.grad,kc.grad),("dv",v.grad,vc.grad)):
        g=g.detach().float().cpu(); bad=~torch.isfinite(g)
        err=(torch.nan_to_num(g)-w.detach()).abs()
        print(D,causal,nm,"nonfinite",int(bad.sum()),"maxerr %.4f"%err.max().item(), "bad idx", bad.nonzero()[:3].tolist(), "err argmax", [int(x) for x in torch.unravel_index(err.argmax(), err.shape)])

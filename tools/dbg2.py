import torch, sys
sys.path.insert(0,'/root/repo')
import flash_attention_softmax_n_amd as pkg
from flash_attention_softmax_n_amd import synth
from oracle.ref_attention import ref_attention_n
dev=torch.device('cuda:0')
D=128
for (B,H,L,S,n,usemask) in ((1,1,64,128,1.0,False),(1,1,64,128,0.0,False),(1,1,256,256,1.0,False),(3,2,200,336,0.0,True),(3,2,200,336,1.0,True)):
    dtype=torch.bfloat16
    q,k,v=(synth.counter_normal(sh,s,dtype=dtype,device=dev).requires_grad_() for sh,s in (((B,H,L,D),1),((B,H,S,D),2),((B,H,S,D),3)))
    do=synth.counter_normal((B,H,L,D),4,std=1.0,dtype=dtype,device=dev)
    gen=torch.Generator().manual_seed(5)
    mask=synth.keypad_mask(B,S,device=dev) if usemask else None
    bias=(1.5*torch.randn(H,L,S,generator=gen)).to(dtype).to(dev)
    out=pkg.flash_attention_n(q,k,v,softmax_n_param=n,attn_mask=mask,attn_bias=bias)
    out.backward(do)
    qc,kc,vc=(t.detach().cpu().float().requires_grad_() for t in (q,k,v))
    o=ref_attention_n(qc,kc,vc,softmax_n_param=n,attn_mask=None if mask is None else mask.cpu(),attn_bias=bias.cpu().float())
    o.backward(do.cpu().float())
    for nm,g,w in (("o",out,o),("dq",q.grad,qc.grad),("dk",k.grad,kc.grad),("dv",v.grad,vc.grad)):
        g=g.detach().float().cpu(); err=(torch.nan_to_num(g)-w.detach()).abs()
        badkeys=(err.amax(dim=(0,1,3))>0.05).nonzero().flatten().tolist() if nm in("dk","dv") else []
        print((B,H,L,S,n,usemask),nm,"maxerr %.4f"%err.max().item(),"max|w| %.3f"%w.abs().max().item(),"bad keys", badkeys[:20], len(badkeys))

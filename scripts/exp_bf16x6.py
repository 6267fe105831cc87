"""CPU emulation of the bf16-split matrix products of the fused MLP kernels (mlp_core_b16.h): an 8 x 256 softplus(100)
network evaluated in fp64 (reference), fp32, and with every operand split into 2 or 3 bf16 planes (each difference exact in
fp32) and the kept cross terms summed.  Per-term sums are formed exactly here, so this is what the SPLIT costs, without
the matrix instruction's own accumulation error.

    python scripts/exp_bf16x6.py
    f32 8.3e-07   x3 1.3e-05   x4 1.2e-05   x6 1.3e-07     (max error / max |value| of the last layer)
"""
import torch
torch.manual_seed(0)
def split(x, n):
    parts=[]; r=x.clone()
    for i in range(n):
        p=r.to(torch.bfloat16).to(torch.float32); parts.append(p); r=r-p
    return parts
def mm_split(W, X, n, terms):
    Ws=split(W,n); Xs=split(X,n)
    acc=torch.zeros(W.shape[0], X.shape[1], dtype=torch.float32)
    # accumulate small terms first? MFMA chain order: as listed
    for (i,j) in terms:
        acc = acc + (Ws[i].double() @ Xs[j].double()).float()   # each product-sum exact-ish then rounded (optimistic on accumulation)
    return acc
def softplus(a): return torch.nn.functional.softplus(a, beta=100)
L=8; Wd=256; P=4096
Ws=[torch.randn(Wd,Wd)*(2**0.5)/Wd**0.5 for _ in range(L)]
x0=torch.randn(Wd,P)*0.5
def run(kind):
    h=x0.clone() if kind!='f64' else x0.double()
    for W in Ws:
        if kind=='f64': a=W.double()@h
        elif kind=='f32': a=W@h
        elif kind=='x3': a=mm_split(W,h,2,[(0,0),(0,1),(1,0)])
        elif kind=='x6': a=mm_split(W,h,3,[(0,2),(2,0),(1,1),(0,1),(1,0),(0,0)])
        elif kind=='x4': a=mm_split(W,h,2,[(1,1),(0,1),(1,0),(0,0)])
        h=softplus(a) if kind!='f64' else torch.nn.functional.softplus(a,beta=100)
    return h.double()
ref=run('f64')
for k in ('f32','x3','x4','x6'):
    o=run(k); print(k, 'max rel err', ((o-ref).abs().max()/ref.abs().max()).item())

"""CPU emulation (torch, no GPU) of the tower arithmetics on C2's shapes against a float64 evaluation of the same
network and coscos2 loss: plain fp32, bf16 x 3 (six products), fp16 x 2 (three products) with power-of-two scales
per tensor, and fp16 x 2 WITHOUT scales -- what decided the design of precision 'f16x2' (DESIGN.md 3.1):

    fp32            emb 3.8e-07  gW 2.0e-05 1.9e-05 1.6e-05 2.5e-05
    bf16x3          emb 2.0e-07  gW 9.0e-06 1.3e-05 7.7e-06 1.2e-05
    f16x2 scaled    emb 4.8e-07  gW 1.8e-05 1.8e-05 1.5e-05 2.6e-05
    f16x2 unscaled  emb 4.5e-07  gW 1.4e-01 1.0e-01 9.2e-02 6.6e-02      (dZ is 1e-6 .. 1e-3: fp16 subnormals)

(errors relative to the tensor's largest entry; the library scales per batch row / weight block / slab, finer than
the per-tensor scales here).    python tools/f16x2_emulation.py"""
import numpy as np, torch
torch.manual_seed(0); np.random.seed(0)
torch.set_num_threads(8)
B=4096; dims=[40,500,500,500,100]
# init like nn.Linear default? reference uses xavier uniform maybe; use nn.Linear default
import torch.nn as nn
lin=[nn.Linear(dims[i],dims[i+1]) for i in range(4)]
for l in lin: nn.init.xavier_uniform_(l.weight)
W=[l.weight.detach().double() for l in lin]; b=[l.bias.detach().double() for l in lin]
x1=torch.randn(B,40).double(); x2=(x1+0.3*torch.randn(B,40)).double()
X=torch.cat([x1,x2]); y=torch.from_numpy(np.random.choice([1,-1],B))

def split_bf16x3(a):
    a=a.float()
    hi=a.view(torch.int32).bitwise_and(-65536).view(torch.float32)
    r=a-hi
    mid=r.view(torch.int32).bitwise_and(-65536).view(torch.float32)
    lo=(r-mid).view(torch.int32).bitwise_and(-65536).view(torch.float32)
    return hi,mid,lo
def mm_bf16x3(a,bm):
    ah,am,al=split_bf16x3(a); bh,bm_,bl=split_bf16x3(bm)
    # smallest first
    acc=al@bh; acc=acc+ah@bl; acc=acc+am@bm_; acc=acc+ah@bm_; acc=acc+am@bh; acc=acc+ah@bh
    return acc
def pow2scale(a,target=14):
    m=float(a.abs().max()); 
    if m==0: return 1.0
    e=np.floor(np.log2(m)); return 2.0**(target-e)
def split_f16x2(a,s):
    a=a.float()*np.float32(s)
    hi=a.half(); r=a-hi.float(); lo=r.half()
    return hi.float(),lo.float()
def mm_f16x2(a,bm,perrow_a=False,scale=True):
    # a [M,K], bm [K,N]
    if scale:
        if perrow_a:
            m=a.abs().amax(dim=1,keepdim=True).clamp_min(1e-30); sa=torch.pow(2.0,14-torch.floor(torch.log2(m))).float()
        else: sa=pow2scale(a)
        sb=pow2scale(bm)
    else: sa=1.0; sb=1.0
    if perrow_a:
        af=a.float()*sa; ah=af.half(); al=(af-ah.float()).half(); ah=ah.float(); al=al.float()
    else: ah,al=split_f16x2(a,sa)
    bh,bl=split_f16x2(bm,sb)
    acc=al@bh; acc=acc+ah@bl; acc=acc+ah@bh
    return acc/ (sa*sb) if not perrow_a else acc/sa/sb
sig=torch.sigmoid
def run(mm,dt):
    # forward
    A=[X.to(dt)]; 
    for l in range(4):
        z=mm(A[-1],W[l].to(dt).t()).to(dt)+b[l].to(dt)
        A.append(sig(z))
    E=A[-1].double()
    e1,e2=E[:B],E[B:]
    # loss grads in f64
    n1=e1.norm(dim=1); n2=e2.norm(dim=1); dot=(e1*e2).sum(1); cs=dot/(n1*n2)
    yy=y
    loss=torch.where(yy==1,(1-cs)/2,cs*cs).sum()
    dcos=torch.where(yy==1,torch.full_like(cs,-0.5),2*cs)
    d1=dcos[:,None]*(e2/(n1*n2)[:,None]-cs[:,None]*e1/(n1*n1)[:,None])
    d2=dcos[:,None]*(e1/(n1*n2)[:,None]-cs[:,None]*e2/(n2*n2)[:,None])
    dA=torch.cat([d1,d2]).to(dt)
    gW=[None]*4; gb=[None]*4
    dZ=(dA*(A[4]*(1-A[4]))).to(dt)
    for l in (3,2,1,0):
        gW[l]=mm(dZ.t().contiguous(),A[l]).to(dt); gb[l]=dZ.sum(0)
        if l>0:
            dA=mm(dZ,W[l].to(dt)).to(dt); dZ=(dA*(A[l]*(1-A[l]))).to(dt)
    return E,float(loss),gW,gb
ref=run(lambda a,b:a@b, torch.float64)
def rel(u,v): return float((u.double()-v.double()).abs().max()/v.double().abs().max())
for name,mm in [('fp32',lambda a,b:a.float()@b.float()),('bf16x3',mm_bf16x3),('f16x2 scaled',mm_f16x2),('f16x2 unscaled',lambda a,b:mm_f16x2(a,b,scale=False))]:
    r=run(mm,torch.float32)
    print('%-15s emb %.2e loss %.2e  gW %s gb %s'%(name,rel(r[0],ref[0]),abs(r[1]-ref[1])/abs(ref[1]),' '.join('%.2e'%rel(r[2][l],ref[2][l]) for l in range(4)),' '.join('%.2e'%rel(r[3][l],ref[3][l]) for l in range(4))))

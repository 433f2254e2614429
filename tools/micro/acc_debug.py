import sys, torch
sys.path.insert(0, ".")
from dcfp_amd import ops
dev=torch.device("cuda:0")
N,C,H,W,Co=2,256,128,256,256
dy=torch.zeros(N,Co,H,W,device=dev); w=torch.randn(Co,C,1,1,device=dev)*0.03
pat=torch.arange(N*C*H*W,device=dev,dtype=torch.float32).reshape(N,C,H,W)
out=pat.clone()
ops.conv2d_dgrad(dy,w,(N,C,H,W),1,0,1,out=out,accumulate=True)
torch.cuda.synchronize()
bad=(out!=pat)
print("mismatches", int(bad.sum()), "of", out.numel())
idx=bad.nonzero()
if len(idx):
    print("first bad", idx[:8].tolist())
    n,c,h,w_=idx[0].tolist()
    print("got", out[n,c,h,w_].item(), "want", pat[n,c,h,w_].item())
    # decode got value as index
    g=int(out[n,c,h,w_].item()); 
    print("got decodes to", (g//(C*H*W), (g//(H*W))%C, (g//W)%H, g%W))
    # histogram over channel index (row m) and pixel
    cs=idx[:,1].unique().tolist(); print("bad channels", cs[:40], len(cs))
    ps=(idx[:,2]*W+idx[:,3]); print("bad pix mod 256 uniq", (ps%256).unique().tolist()[:40])
    print("bad n", idx[:,0].unique().tolist())

"""Which parameter slices of the ResNet-18 curvature product differ between two calls?"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import pytorchhessianfree_amd as hf
from pytorchhessianfree_amd import _lib, curvature, modelprep, testproblems as tp
hf.configure()
dev = "cuda"
model, (x, t), lossf = tp.resnet18_mnist(32, device=dev)
modelprep.prepare_model(model, channels_last=True)
params = [p for p in model.parameters() if p.requires_grad]
names = [n for n, p in model.named_parameters() if p.requires_grad]
out = model(x)
op = curvature.GGNOperator(lossf(out, t), out, params)
v = torch.randn(op.n, device=dev, generator=torch.Generator(device=dev).manual_seed(1))
ref = op(v).clone()
for rep in range(4):
    got = op(v)
    if torch.equal(got, ref):
        print("rep", rep, "identical"); continue
    off = 0
    for n_, p in zip(names, params):
        a, b = ref[off:off + p.numel()], got[off:off + p.numel()]
        if not torch.equal(a, b):
            print("rep", rep, n_, tuple(p.shape), "maxdiff", float((a - b).abs().max()), "of", float(a.abs().max()))
        off += p.numel()
# kernel-level stress on the shared scratch
def cl(t): return t.contiguous(memory_format=torch.channels_last)
shapes = [(32,7,7,64,64,3,1,1),(32,7,7,64,128,3,2,1),(32,4,4,128,128,3,1,1),(32,2,2,256,256,3,1,1),(32,2,2,256,512,3,2,1),(32,1,1,512,512,3,1,1)]
state = []
for (n,h,w,c,k,r,st,pd) in shapes:
    oh = (h + 2*pd - r)//st + 1
    x_, w_ = cl(torch.randn(n,c,h,w,device=dev)), cl(torch.randn(k,c,r,r,device=dev))
    gy = cl(torch.randn(n,k,oh,oh,device=dev)); wT = w_.permute(1,2,3,0).contiguous()
    state.append((n,h,w,c,k,r,st,pd,oh,x_,w_,gy,wT))
def run(i):
    n,h,w,c,k,r,st,pd,oh,x_,w_,gy,wT = state[i]
    y = cl(torch.empty(n,k,oh,oh,device=dev)); gx = torch.empty_like(x_); gw = torch.zeros_like(w_)
    _lib.conv2d_nhwc(0, y, x_, w_, n,h,w,c,k,r,r,(st,st),(pd,pd))
    _lib.conv2d_nhwc_backward(gx, gw, gy, x_, wT, n,h,w,c,k,r,r,(st,st),(pd,pd))
    return y, gx, gw
refs = [run(i) for i in range(len(state))]
bad = 0
for rep in range(30):
    for i in torch.randperm(len(state)).tolist():
        res = run(i)
        for nm, a, b in zip("y gx gw".split(), refs[i], res):
            if not torch.equal(a, b):
                bad += 1
                print("kernel stress: shape", shapes[i], nm, "differs", float((a-b).abs().max()))
print("kernel stress mismatches:", bad)

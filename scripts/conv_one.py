"""One convolution geometry, the three slab-mode launches, N times each (for rocprofv3 counters).
    python3 scripts/conv_one.py n h w c k r stride pad [reps]"""
import os, sys
sys.path.insert(0, os.getcwd())
import torch
from pytorchhessianfree_amd import _lib

n, h, w, c, k, r, st, pd = [int(a) for a in sys.argv[1:9]]
reps = int(sys.argv[9]) if len(sys.argv) > 9 else 10
stride, pad = (st, st), (pd, pd)
oh = (h + 2 * pd - r) // st + 1
cl = lambda t: t.contiguous(memory_format=torch.channels_last)
DEV = "cuda"
x2, w2 = cl(torch.randn(n, 2 * c, h, w, device=DEV)), cl(torch.randn(k, 2 * c, r, r, device=DEV))
x, wt = cl(torch.randn(n, c, h, w, device=DEV)), cl(torch.randn(k, c, r, r, device=DEV))
gy = cl(torch.randn(n, k, oh, oh, device=DEV))
wT = wt.permute(1, 2, 3, 0).contiguous()
sT = _lib.conv_plan(0, n, h, w, 2 * c, k, r, r, stride, pad)
sD = _lib.conv_plan(1, n, h, w, c, k, r, r, stride, pad)
sW = _lib.conv_plan(2, n, h, w, c, k, r, r, stride, pad)
ys = torch.empty(sT, n * oh * oh * k, device=DEV)
gxs = torch.empty(sD, x.numel(), device=DEV)
gws = torch.zeros(sW, wt.numel(), device=DEV)
print("splits", sT, sD, sW, file=sys.stderr)
for _ in range(reps):
    _lib.conv2d_nhwc_slabs(0, ys, x2, w2, n, h, w, 2 * c, k, r, r, stride, pad, sT)
    _lib.conv2d_nhwc_slabs(1, gxs, gy, wT, n, h, w, c, k, r, r, stride, pad, sD)
    _lib.conv2d_nhwc_slabs(2, gws, x, gy, n, h, w, c, k, r, r, stride, pad, sW)
torch.cuda.synchronize()

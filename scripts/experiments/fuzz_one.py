import sys, os, warnings
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import pytorchhessianfree_amd as hf
from oracle import pcg as oracle
case, n, dtype = 1393, 64, torch.float64
g = torch.Generator().manual_seed(case)
d = (torch.rand(n, generator=g) * 10 + 0.05).to(dtype); b = torch.randn(n, generator=g).to(dtype)
x0 = torch.randn(n, generator=g).to(dtype); diag = torch.rand(n, generator=g).to(dtype)
dd = d.cuda()
Mg = hf.DiagonalPreconditioner(diag.cuda(), 0.1); minv = Mg.minv.cpu()
kw = dict(max_iter=60, tol=0.01, atol=1e-6, martens_conv_crit=False, store_x_at_iters=list(range(61)))
ox, _, orr = oracle.pcg(lambda v: d * v + 0.0 * v, b, x0=x0, M=lambda v: minv * v, accumulate="fp64", **kw)
gx, _, grr = hf.cg(hf.DampedCurvature(lambda v: dd * v, 0.0), b.cuda(), x0=x0.cuda(), M=lambda v: Mg.minv * v, **kw)
# also oracle in reference mode
rx, _, _ = oracle.pcg(lambda v: d * v + 0.0 * v, b, x0=x0, M=lambda v: minv * v, **kw)
print(orr, grr, len(ox), len(gx), len(rx))
for i in range(min(len(ox), len(gx), len(rx))):
    e = float((gx[i].cpu() - ox[i]).abs().max() / ox[i].abs().max()); e2 = float((rx[i] - ox[i]).abs().max() / ox[i].abs().max())
    print(i, "gpu-vs-oracle64 %.2e   oracleRef-vs-oracle64 %.2e" % (e, e2))

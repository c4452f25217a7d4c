import sys, os, warnings
sys.path.insert(0, os.getcwd())
import torch
import pytorchhessianfree_amd as hf
from pytorchhessianfree_amd import curvature, testproblems as tp, modelprep
order = sys.argv[1]  # "graph_first" or "eager_first"
gm, (gx_, gt_), lossf = tp.resnet18_mnist(batch_size=32, device="cuda", data_seed=1000)
modelprep.prepare_model(gm, channels_last=True)
gp = list(gm.parameters())
def builder():
    o = gm(gx_); return curvature.GGNOperator(lossf(o, gt_), o, gp)
v = torch.randn(sum(p.numel() for p in gp), device="cuda", generator=torch.Generator(device="cuda").manual_seed(0))
# stock NCHW model as reference (separate module, no fusions)
sm, (sx, st), _ = tp.resnet18_mnist(batch_size=32, device="cuda", data_seed=1000)
sp = list(sm.parameters()); so = sm(sx); ref = curvature.GGNOperator(lossf(so, st), so, sp)(v).clone()
def rel(a): return float((a - ref).abs().max() / ref.abs().max())
if order == "eager_first":
    e = builder(); r_e = e(v).clone(); r_e2 = e(v).clone(); del e
    g = curvature.GraphedOperator(builder, params=gp); r_g = g(v).clone(); r_g2 = g(v).clone()
else:
    g = curvature.GraphedOperator(builder, params=gp); r_g = g(v).clone(); r_g2 = g(v).clone(); del g
    e = builder(); r_e = e(v).clone(); r_e2 = e(v).clone()
print("ORDER", order, "eager err %.1e %.1e | graph err %.1e %.1e" % (rel(r_e), rel(r_e2), rel(r_g), rel(r_g2)))

import sys, os, warnings, time
sys.path.insert(0, os.getcwd())
import torch, numpy as np
import pytorchhessianfree_amd as hf
from pytorchhessianfree_amd import curvature, testproblems as tp, modelprep
from pytorchhessianfree_amd.utils import vector_to_parameter_list
from oracle import pcg as oracle, backpack_restated as bp
lam, iters = float(sys.argv[1]), int(sys.argv[2])
# CPU oracle: stock model, BackPACK's algorithm, reference-order PCG
model, (x, t), lossf = tp.resnet18_mnist(batch_size=32, device="cpu", data_seed=1000)
params = [p for p in model.parameters()]
out = model(x); loss = lossf(out, t)
grad = torch.cat([g.reshape(-1) for g in torch.autograd.grad(loss, params, retain_graph=True)])
def mvp(v):
    return torch.cat([g.reshape(-1) for g in bp.ggn_vector_product_from_plist(loss, out, params, vector_to_parameter_list(v, params))]).detach()
t0 = time.time()
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    ox, om, oreason = oracle.pcg(lambda v: mvp(v) + lam * v, -grad, max_iter=iters, martens_conv_crit=True, store_x_at_iters=None)
print("cpu oracle", len(ox) - 1, oreason, "%.1fs" % (time.time() - t0))
# GPU product: fused layers, hipGraph
gm, (gx_, gt_), _ = tp.resnet18_mnist(batch_size=32, device="cuda", data_seed=1000)
modelprep.prepare_model(gm)
gp = [p for p in gm.parameters()]
ggrad = curvature.flatten_into(torch.autograd.grad(lossf(gm(gx_), gt_), gp), gp)
print("grad rel diff", float((ggrad.cpu() - grad).norm() / grad.norm()))
def builder():
    o = gm(gx_); return curvature.GGNOperator(lossf(o, gt_), o, gp)
op = curvature.maybe_graphed(builder, params=gp)
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    gx, gmm, greason = hf.cg(hf.DampedCurvature(op, lam), -ggrad, max_iter=iters, martens_conv_crit=True, store_x_at_iters=None)
print("gpu", len(gx) - 1, greason)
k = min(len(gx), len(ox))
for i in range(k):
    if gx[i] is not None and ox[i] is not None and i > 0:
        a, b = gx[i].cpu(), ox[i]
        print("it %3d  rel l2 %.2e  maxrel %.2e | m gpu %.6e cpu %.6e rel %.1e" % (i, float((a - b).norm() / b.norm()), float((a - b).abs().max() / b.abs().max()), float(gmm[i]), float(om[i]), abs(float(gmm[i]) - float(om[i])) / abs(float(om[i]))))

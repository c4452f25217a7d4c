import sys, os, warnings, time
sys.path.insert(0, os.getcwd())
import torch
import pytorchhessianfree_amd as hf
from pytorchhessianfree_amd import curvature, testproblems as tp, modelprep
burn = int(sys.argv[1]); cl = int(sys.argv[2])
if burn:
    a = torch.randn(3000, 3000)
    t0 = time.time()
    while time.time() - t0 < burn: a = (a @ a).tanh()
for trial in range(2):
    gm, (gx_, gt_), lossf = tp.resnet18_mnist(batch_size=32, device="cuda", data_seed=1000)
    modelprep.prepare_model(gm, channels_last=bool(cl))
    gp = list(gm.parameters())
    ggrad = curvature.flatten_into(torch.autograd.grad(lossf(gm(gx_), gt_), gp), gp)
    def builder():
        o = gm(gx_); return curvature.GGNOperator(lossf(o, gt_), o, gp)
    op = curvature.maybe_graphed(builder, params=gp)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        gx, gmm, greason = hf.cg(hf.DampedCurvature(op, 1e-3), -ggrad, max_iter=80, martens_conv_crit=True, store_x_at_iters=None)
    print("trial", trial, "iters", len(gx) - 1, greason, "m_end %.6f" % float(gmm[-1]), "m13 %.6f" % float(gmm[13]), getattr(op, "mode", "eager")[:10], flush=True)
    del op

import sys, os
sys.path.insert(0, os.getcwd())
import torch
import pytorchhessianfree_amd as hf
from pytorchhessianfree_amd import curvature, testproblems as tp, modelprep
def product(batch, prep, cl):
    m, (x, t), lf = tp.resnet18_mnist(batch_size=batch, device="cuda", data_seed=1000)
    if prep == "all": modelprep.prepare_model(m, channels_last=cl)
    elif prep == "bn": modelprep.fuse_eval_batchnorm(m)
    elif prep == "conv": modelprep.fuse_conv_tangent(m, channels_last=cl)
    elif prep == "bnconv": modelprep.fuse_eval_batchnorm(m); modelprep.fuse_conv_tangent(m, channels_last=cl)
    ps = list(m.parameters())
    o = m(x); op = curvature.GGNOperator(lf(o, t), o, ps)
    v = torch.randn(op.n, device="cuda", generator=torch.Generator(device="cuda").manual_seed(0))
    return op(v).clone()
for batch in (8, 32):
    ref = product(batch, "none", False)
    for prep, cl in [("all", False), ("conv", True), ("bnconv", True), ("all", True)]:
        r = product(batch, prep, cl)
        print("batch", batch, prep, "cl" if cl else "nchw", "rel err %.2e" % float((r - ref).abs().max() / ref.abs().max()))

import sys, os, warnings
sys.path.insert(0, os.getcwd())
import torch
import pytorchhessianfree_amd as hf
from pytorchhessianfree_amd import curvature, testproblems as tp, modelprep
cl = int(sys.argv[1]) if len(sys.argv) > 1 else 1
m, (x, t), lf = tp.resnet18_mnist(batch_size=32, device="cuda", data_seed=1000)
modelprep.prepare_model(m, channels_last=bool(cl))
ps = list(m.parameters())
grad = curvature.flatten_into(torch.autograd.grad(lf(m(x), t), ps), ps)
def builder():
    o = m(x); return curvature.GGNOperator(lf(o, t), o, ps)
for trial in range(4):
    eager = builder()
    gen = torch.Generator(device="cuda").manual_seed(trial)
    vs = [torch.randn(eager.n, device="cuda", generator=gen) for _ in range(6)]
    refs = [eager(v).clone() for v in vs]
    del eager
    g = curvature.GraphedOperator(builder, params=ps)
    errs = []
    for rep in range(3):
        for v, r in zip(vs, refs):
            out = g(v)
            errs.append(float((out - r).abs().max() / r.abs().max()))
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        xs, ms, reason = hf.cg(hf.DampedCurvature(g, 1e-3), -grad, max_iter=40, martens_conv_crit=True, store_x_at_iters=None)
    print("trial", trial, "max graph-vs-eager err %.2e" % max(errs), "| cg", len(xs) - 1, reason, flush=True)
    del g

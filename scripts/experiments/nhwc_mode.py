"""NHWC product in a fresh process: error against a stock-autograd product and time per
product, with MIOpen in find mode (benchmark=1) or immediate mode (benchmark=0)."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
import pytorchhessianfree_amd as hf
from pytorchhessianfree_amd import curvature, modelprep, testproblems as tp

bench = int(sys.argv[1]); cl = int(sys.argv[2]) if len(sys.argv) > 2 else 1
torch.backends.cudnn.benchmark = bool(bench)
dev = torch.device("cuda", 0)

def problem():
    return tp.resnet18_mnist(batch_size=32, seed=0, device=dev, data_seed=1000)

model, (x, t), lossf = problem()
modelprep.prepare_model(model, channels_last=bool(cl))
params = [p for p in model.parameters() if p.requires_grad]
n = sum(p.numel() for p in params)
def builder():
    out = model(x)
    return curvature.GGNOperator(lossf(out, t), out, params)
op = curvature.GraphedOperator(builder, params=params)
v = torch.randn(n, device=dev, generator=torch.Generator(device=dev).manual_seed(7))
got = op(v).clone()
m2, (x2, t2), l2 = problem()
p2 = [p for p in m2.parameters() if p.requires_grad]
o2 = m2(x2)
want = curvature.GGNOperator(l2(o2, t2), o2, p2)(v).clone()
err = float((got - want).abs().max() / want.abs().max())
got2 = op(v).clone()
err2 = float((got2 - want).abs().max() / want.abs().max())
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(100): op(v)
torch.cuda.synchronize(); ms = (time.perf_counter() - t0) * 10
print("RESULT bench=%d cl=%d err %.2e err2 %.2e  %.3f ms/product" % (bench, cl, err, err2, ms), flush=True)

"""bench.py's NHWC check with diagnostics: where is a bad product wrong, is a second replay
wrong too, is an eager NHWC product wrong too?  (run under rocprofv3, where the check fails
in about half of the processes)"""
import os, sys
sys.path.insert(0, os.getcwd())
import torch
import pytorchhessianfree_amd as hf
from pytorchhessianfree_amd import curvature, modelprep, testproblems as tp

dev = torch.device("cuda", 0)
order = sys.argv[1] if len(sys.argv) > 1 else "stock_first"
with_grad = int(sys.argv[2]) if len(sys.argv) > 2 else 1
workload = sys.argv[3] if len(sys.argv) > 3 else "resnet18"

def problem():
    make = {"resnet18": tp.resnet18_mnist, "allcnnc": tp.allcnnc_cifar100, "resnet50": tp.resnet50_small_images}[workload]
    return make(batch_size=32, seed=0, device=dev, data_seed=1000)

model, (x, t), lossf = problem()
modelprep.prepare_model(model, channels_last=True)
params = [p for p in model.parameters() if p.requires_grad]
names = [n for n, p in model.named_parameters() if p.requires_grad]
n = sum(p.numel() for p in params)
if with_grad:
    grad = curvature.flatten_into(torch.autograd.grad(lossf(model(x), t), params), params)
def builder():
    out = model(x)
    return curvature.GGNOperator(lossf(out, t), out, params)
op = curvature.GraphedOperator(builder, params=params)
v = torch.randn(n, device=dev, generator=torch.Generator(device=dev).manual_seed(7))

def stock():
    m2, (x2, t2), l2 = problem()
    p2 = [p for p in m2.parameters() if p.requires_grad]
    torch.backends.cudnn.benchmark = False
    o2 = m2(x2)
    r = curvature.GGNOperator(l2(o2, t2), o2, p2)(v).clone()
    torch.backends.cudnn.benchmark = True
    return r

if order == "stock_first":
    want = stock(); got = op(v).clone()
else:
    got = op(v).clone(); want = stock()
torch.cuda.synchronize()
scale = float(want.abs().max())
err = float((got - want).abs().max()) / scale
msg = "RESULT order=%s grad=%d err %.2e" % (order, with_grad, err)
if not err < 1e-4:
    got2 = op(v).clone()
    e2 = float((got2 - want).abs().max()) / scale
    eager = builder()
    got3 = eager(v).clone()
    e3 = float((got3 - want).abs().max()) / scale
    got4 = op(v).clone()
    e4 = float((got4 - want).abs().max()) / scale
    bad, off = [], 0
    for name, p in zip(names, params):
        seg = slice(off, off + p.numel()); off += p.numel()
        e = float((got[seg] - want[seg]).abs().max()) / scale
        if not e < 1e-4:
            bad.append("%s:%.1e" % (name, e))
    shapes = {nm: tuple(p.shape) for nm, p in zip(names, params)}
    msg += " | shapes " + " ".join("%s%s" % (b.split(":")[0], shapes[b.split(":")[0]]) for b in bad[:12])
    msg += " | replay2 %.2e eager %.2e replay3 %.2e | nan %d inf %d | bad params %d/%d: %s" % (
        e2, e3, e4, int(torch.isnan(got).sum()), int(torch.isinf(got).sum()), len(bad), len(names), " ".join(bad[:12]))
print(msg, flush=True)

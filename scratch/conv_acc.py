import sys, os
sys.path.insert(0, os.getcwd())
import torch
from pytorchhessianfree_amd import curvature, testproblems as tp, modelprep
def grads(device, dtype, prep=False, bench=False):
    torch.backends.cudnn.benchmark = bench
    m, (x, t), lf = tp.resnet18_mnist(batch_size=32, device="cpu", data_seed=1000)
    m = m.to(device, dtype); x = x.to(device, dtype); t = t.to(device)
    if prep: modelprep.prepare_model(m)
    ps = list(m.parameters())
    o = m(x); l = lf(o, t)
    g = torch.autograd.grad(l, ps)
    return o.detach().cpu().double(), torch.cat([a.reshape(-1) for a in g]).cpu().double(), float(l)
def rel(a, b): return float((a - b).norm() / b.norm())
o64, g64, l64 = grads("cpu", torch.float64)
for name, args in [("gpu f32 stock bench", ("cuda", torch.float32, False, True)), ("gpu f32 prepared bench", ("cuda", torch.float32, True, True))]:
    o, g, l = grads(*args)
    print("%-22s out rel %.2e  grad rel %.2e  loss diff %.2e" % (name, rel(o, o64), rel(g, g64), abs(l - l64)))

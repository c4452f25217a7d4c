import sys, os
sys.path.insert(0, os.getcwd())
import torch
from pytorchhessianfree_amd import testproblems as tp
def grads(device, dtype):
    m, (x, t), lf = tp.resnet18_mnist(batch_size=32, device="cpu", data_seed=1000)
    m = m.to(device, dtype); x = x.to(device, dtype); t = t.to(device)
    ps = list(m.parameters()); names = [n for n, _ in m.named_parameters()]
    g = torch.autograd.grad(lf(m(x), t), ps)
    return names, [a.cpu().double() for a in g]
names, g64 = grads("cpu", torch.float64)
_, gg = grads("cuda", torch.float32)
_, gc = grads("cpu", torch.float32)
for n, a, b, c in zip(names, gg, g64, gc):
    r = float((a - b).norm() / b.norm()); rc = float((c - b).norm() / b.norm())
    if r > 2e-6: print("%-32s shape %-22s gpu rel %.2e cpu rel %.2e norm %.3e" % (n, tuple(a.shape), r, rc, float(b.norm())))

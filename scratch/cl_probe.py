import sys, os, collections, types
sys.path.insert(0, os.getcwd())
import torch
from torch.profiler import profile, ProfilerActivity
from pytorchhessianfree_amd import curvature, testproblems as tp
cl = int(sys.argv[1])
def aff(self, x):
    s = self.weight * torch.rsqrt(self.running_var + self.eps); t = self.bias - self.running_mean * s
    return x * s.view(1, -1, 1, 1) + t.view(1, -1, 1, 1)
m, (x, t), lf = tp.resnet18_mnist(batch_size=32, device="cuda", data_seed=1000)
for mod in m.modules():
    if isinstance(mod, torch.nn.BatchNorm2d): mod.forward = types.MethodType(aff, mod)
if cl:
    m = m.to(memory_format=torch.channels_last); x = x.contiguous(memory_format=torch.channels_last)
ps = [p for p in m.parameters()]
o = m(x); op = curvature.GGNOperator(lf(o, t), o, ps)
v = torch.randn(op.n, device="cuda")
for _ in range(3): op(v)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    op(v); torch.cuda.synchronize()
c = collections.Counter(); tm = collections.Counter()
for e in prof.events():
    for k in e.kernels:
        n = k.name[:40]; c[n] += 1; tm[n] += k.duration
print("cl", cl, "total kernels", sum(c.values()), "device us", sum(tm.values()))
for n, k in c.most_common(14): print("  %4d %8.1f %s" % (k, tm[n], n))

import sys, os, collections, types, time
sys.path.insert(0, os.getcwd())
import torch
from torch.profiler import profile, ProfilerActivity
from pytorchhessianfree_amd import curvature, testproblems as tp, modelprep
cl = int(sys.argv[1])
torch.backends.cudnn.benchmark = True
def aff(self, x):
    s = self.weight * torch.rsqrt(self.running_var + self.eps); t = self.bias - self.running_mean * s
    return x * s.view(1, -1, 1, 1) + t.view(1, -1, 1, 1)
m, (x, t), lf = tp.resnet18_mnist(batch_size=32, device="cuda", data_seed=1000)
for mod in m.modules():
    if isinstance(mod, torch.nn.BatchNorm2d): mod.forward = types.MethodType(aff, mod)
modelprep.fuse_conv_tangent(m)
if cl:
    m = m.to(memory_format=torch.channels_last); x = x.contiguous(memory_format=torch.channels_last)
ps = [p for p in m.parameters()]
def builder():
    o = m(x); return curvature.GGNOperator(lf(o, t), o, ps)
op = builder()
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
for n, k in c.most_common(12): print("  %4d %8.1f %s" % (k, tm[n], n))
del op
import gc; gc.collect()
g = curvature.GraphedOperator(builder, params=ps)
g.input_buffer.copy_(v)
for _ in range(5): g(g.input_buffer)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(100): g(g.input_buffer)
torch.cuda.synchronize(); print("cl", cl, "graph ms/matvec %.3f" % ((time.perf_counter() - t0) * 10))

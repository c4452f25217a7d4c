import sys, os, collections
sys.path.insert(0, os.getcwd())
import torch
from torch.profiler import profile, ProfilerActivity
from pytorchhessianfree_amd import curvature, testproblems as tp, modelprep
m, (x, t), lf = tp.resnet18_mnist(batch_size=32, device="cuda", data_seed=1000)
modelprep.fuse_eval_batchnorm(m)
ps = [p for p in m.parameters()]
o = m(x); op = curvature.GGNOperator(lf(o, t), o, ps)
v = torch.randn(op.n, device="cuda")
for _ in range(3): op(v)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    op(v); torch.cuda.synchronize()
evs = prof.events()
# attribute each kernel launch (cpu op with device kernels) to enclosing evaluate_function
tops = [e for e in evs if e.name.startswith("autograd::engine::evaluate_function")]
cnt = collections.Counter(); tim = collections.Counter(); kcnt = collections.Counter()
def kernels_under(e):
    n = 0; t = 0.0
    stack = [e]
    while stack:
        c = stack.pop()
        for k in c.kernels: n += 1; t += k.duration
        stack.extend(c.cpu_children)
    return n, t
for e in tops:
    name = e.name.split(": ")[-1]
    n, t_ = kernels_under(e)
    cnt[name] += 1; kcnt[name] += n; tim[name] += t_
print("node, calls, kernels, device_us")
for name, _ in sorted(kcnt.items(), key=lambda kv: -kv[1])[:30]:
    print("%-45s %4d %5d %9.1f" % (name, cnt[name], kcnt[name], tim[name]))
print("total kernels", sum(kcnt.values()))

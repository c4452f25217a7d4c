import sys, os, collections
sys.path.insert(0, os.getcwd())
import torch
from torch.profiler import profile, ProfilerActivity
import pytorchhessianfree_amd as hf
from pytorchhessianfree_amd import curvature, testproblems as tp, modelprep
m, (x, t), lf = tp.resnet18_mnist(batch_size=32, device="cuda", data_seed=1000)
modelprep.prepare_model(m)
ps = list(m.parameters())
o = m(x); op = curvature.GGNOperator(lf(o, t), o, ps)
v = torch.randn(op.n, device="cuda")
for _ in range(3): op(v)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    op(v); torch.cuda.synchronize()
c = collections.Counter(); tm = collections.Counter()
for e in prof.events():
    for k in e.kernels:
        n = k.name[:60]; c[n] += 1; tm[n] += k.duration
print("total kernels", sum(c.values()), "device us", sum(tm.values()))
for n, k in sorted(c.items(), key=lambda kv: -tm[kv[0]])[:28]: print("  %4d %8.1f us  %s" % (k, tm[n], n))
tops = [e for e in prof.events() if e.name.startswith("autograd::engine::evaluate_function")]
per = collections.Counter(); cnt = collections.Counter()
def walk(e):
    n = len(e.kernels)
    for ch in e.cpu_children: n += walk(ch)
    return n
for e in tops:
    nm = e.name.split(": ")[-1]; per[nm] += walk(e); cnt[nm] += 1
print(sorted(((v, cnt[k], k) for k, v in per.items()), reverse=True)[:14])

import sys, os, collections
sys.path.insert(0, os.getcwd())
import torch
from torch.profiler import profile, ProfilerActivity
from pytorchhessianfree_amd import curvature, testproblems as tp, modelprep
torch.backends.cudnn.benchmark = True
os.environ.setdefault("MIOPEN_USER_DB_PATH", os.path.join(os.getcwd(), "pytorchhessianfree_amd", "miopen_db"))
m, (x, t), lf = tp.resnet18_mnist(batch_size=32, device="cuda", data_seed=1000)
modelprep.prepare_model(m)
ps = [p for p in m.parameters()]
o = m(x); op = curvature.GGNOperator(lf(o, t), o, ps)
v = torch.randn(op.n, device="cuda")
for _ in range(3): op(v)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    op(v); torch.cuda.synchronize()
evs = prof.events()
tops = [e for e in evs if e.name.startswith("autograd::engine::evaluate_function")]
per = collections.defaultdict(collections.Counter); calls = collections.Counter()
def walk(e, acc):
    for k in e.kernels: acc[k.name[:48]] += 1
    for c in e.cpu_children: walk(c, acc)
for e in tops:
    name = e.name.split(": ")[-1]; calls[name] += 1; walk(e, per[name])
print("nodes:", sorted(((sum(per[n].values()), calls[n], n) for n in per), reverse=True)[:12])
for name in ("_ConvBwdBackward", "ConvolutionBackward0", "_ConvBackward"):
    print(name, "calls", calls[name], "kernels", sum(per[name].values()))
    for k, c in per[name].most_common(14): print("   %4d %s" % (c, k))
# kernels not under any evaluate_function
allk = sum(len(e.kernels) for e in evs)
print("all kernel records", allk)

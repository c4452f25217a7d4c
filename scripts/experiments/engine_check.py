"""Build the fused curvature engine for the ResNet-18 workload, check it against the autograd
product and the float64 stock product, time graph replays."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("HF_ENGINE_DEBUG", "1")
import pytorchhessianfree_amd as hf
from pytorchhessianfree_amd import curvature, modelprep, testproblems as tp
from pytorchhessianfree_amd.engine import FusedGGNEngine
hf.configure()
dev = "cuda"
model, (x, t), lossf = tp.resnet18_mnist(32, device=dev)
modelprep.prepare_model(model, channels_last=True)
params = [p for p in model.parameters() if p.requires_grad]

def builder():
    out = model(x)
    return curvature.ggn_operator(lossf(out, t), out, params)

op = builder()
print("operator:", type(op).__name__)
v = torch.randn(op.n, device=dev, generator=torch.Generator(device=dev).manual_seed(1))
got = op(v).clone()
print("repeatable:", bool(torch.equal(op(v), got)))
os.environ["HF_ENGINE"] = "0"
ref32 = builder()(v).clone()
os.environ["HF_ENGINE"] = "1"
m64, (x64, t64), _ = tp.resnet18_mnist(32, device=dev)
m64 = m64.double(); p64 = [p for p in m64.parameters()]
o64 = m64(x64.double())
want = curvature.GGNOperator(lossf(o64, t64), o64, p64)(v.double())
sc = want.abs().max()
print("engine vs float64:", float((got.double() - want).abs().max() / sc), " autograd fp32 vs float64:", float((ref32.double() - want).abs().max() / sc))
names = [n for n, p in model.named_parameters()]
off = 0
worst = []
for n_, p in zip(names, params):
    a, b = got[off:off+p.numel()].double(), want[off:off+p.numel()]
    worst.append((float((a-b).abs().max() / sc), n_)); off += p.numel()
print("worst params:", sorted(worst, reverse=True)[:4])
if type(op).__name__ == "FusedGGNEngine":
    from pytorchhessianfree_amd import _lib
    tensors, perms, splits = op._pack_args()
    tensors = [t if t is not None else torch.zeros(p.numel(), device=dev) for t, p in zip(tensors, params)]
    outv = torch.empty(op.n, device=dev)
    def timed(fn, reps=20):
        fn(); torch.cuda.synchronize()
        gg = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gg):
            for _ in range(reps): fn()
        gg.replay(); torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5): gg.replay()
        torch.cuda.synchronize(); return (time.perf_counter() - t0) / (5 * reps) * 1e6
    print("pack as engine: %.1f us" % timed(lambda: _lib.pack_ex(outv, tensors, perms, splits)))
    print("pack no splits: %.1f us" % timed(lambda: _lib.pack_ex(outv, tensors, perms, {})))
    print("pack no perms no splits: %.1f us" % timed(lambda: _lib.pack_ex(outv, tensors, {}, {})))
    big = {i for i, t in enumerate(tensors) if t.numel() > 1_000_000}
    print("pack perms only on >1M tensors: %.1f us" % timed(lambda: _lib.pack_ex(outv, tensors, {i: perms[i] for i in perms if i in big}, {})))
    print("splits:", {i: splits[i][0] for i in splits})
del op
g = curvature.GraphedOperator(builder, params=params)
print("graphed:", type(g.op).__name__)
gv = g(v).clone()
print("graph vs eager:", float((gv - got).abs().max() / got.abs().max()))
torch.cuda.synchronize()
for reps in (50, 200):
    t0 = time.perf_counter()
    for _ in range(reps):
        g.replay_local()
    torch.cuda.synchronize()
    print(f"replay: {(time.perf_counter() - t0) / reps * 1e3:.3f} ms per product")

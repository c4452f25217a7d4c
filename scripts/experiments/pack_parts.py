"""Where does hf_pack_ex spend its time on the ResNet-18 product?  Times the gather for subsets of the
engine's source tensors (split-K weight-gradient slabs / tensors with dead taps / the rest)."""
import os, sys
sys.path.insert(0, os.getcwd())
import torch
import pytorchhessianfree_amd as hf
from pytorchhessianfree_amd import _lib, curvature, modelprep
from pytorchhessianfree_amd import testproblems as tp

hf.configure()
model, (x, t), lossf = tp.resnet18_mnist(batch_size=32, device="cuda", data_seed=tp.RESNET18_B32_SEPARATED_SEEDS[0])
modelprep.prepare_model(model, channels_last=True)
params = [p for p in model.parameters() if p.requires_grad]
out = model(x)
op = curvature.ggn_operator(lossf(out, t), out, params)
eng = getattr(op, "engine", None) or getattr(op, "op", op)
tensors, perms, splits = eng._pack_args()
tensors = list(tensors)
for i, t_ in enumerate(tensors):  # (the classifier's gradients come from the head kernel: stand-ins of the right size)
    if t_ is None:
        tensors[i] = torch.zeros(params[i].numel(), device="cuda")
live = eng._pack_live
v = torch.randn(eng.n, device="cuda")
op.local(v)


def timed(idxs, label):
    ts = [tensors[i] for i in idxs]
    pm = {k: perms[i] for k, i in enumerate(idxs) if i in perms}
    sp = {k: splits[i] for k, i in enumerate(idxs) if i in splits}
    lv = {k: live[i] for k, i in enumerate(idxs) if i in live}
    n = sum(t_.numel() for t_ in ts)
    dst = torch.empty(n, device="cuda")
    rd = sum(t_.numel() * splits.get(i, (1, 0))[0] * (bin(live[i]).count("1") / perms[i][1] if i in live else 1.0)
             for t_, i in zip(ts, idxs))
    for _ in range(3):
        _lib.pack_ex(dst, ts, pm, sp, 1.0, lv)
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    g = torch.cuda.CUDAGraph()
    inner = 10
    with torch.cuda.stream(side):
        with torch.cuda.graph(g, stream=side):  # (eager launches are bound by the host's argument marshalling)
            for _ in range(inner):
                _lib.pack_ex(dst, ts, pm, sp, 1.0, lv)
    torch.cuda.synchronize()
    g.replay()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    reps = 20
    for _ in range(reps):
        g.replay()
    b.record()
    torch.cuda.synchronize()
    us = a.elapsed_time(b) * 1e3 / (reps * inner)
    mb = 4e-6 * (n + rd)
    print(f"{label:38s} {len(idxs):3d} tensors  {n/1e6:7.3f} M out  {mb:7.1f} MB  {us:7.2f} us  {mb/us*1e-3:6.2f} TB/s")


allidx = list(range(len(tensors)))
timed(allidx, "all")
timed([i for i in allidx if i in live], "dead-tap tensors (layer4 3x3)")
timed([i for i in allidx if i in splits and i in perms and i not in live], "split + permuted (3x3, layers 1-3)")
timed([i for i in allidx if i in splits and i not in perms], "split, not permuted (1x1, bn sums)")
timed([i for i in allidx if i not in splits and i not in live], "neither split nor dead taps")
timed([i for i in allidx if i not in live], "everything but the dead-tap tensors")
for i in allidx:
    if tensors[i].numel() >= 100000:
        timed([i], f"  #{i} {tuple(params[i].shape)} s={splits.get(i, (1,))[0]} live={live.get(i, 0):b}")

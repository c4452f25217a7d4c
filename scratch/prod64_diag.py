"""GGN product: stock fp32, prepared NCHW, prepared NHWC (eager) against a float64 stock product"""
import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
import pytorchhessianfree_amd as hf
from pytorchhessianfree_amd import curvature, modelprep, testproblems as tp
dev = torch.device("cuda", 0)
workload = sys.argv[1]
make = {"resnet18": tp.resnet18_mnist, "allcnnc": tp.allcnnc_cifar100, "resnet50": tp.resnet50_small_images}[workload]
def problem():
    return make(batch_size=32, seed=0, device=dev, data_seed=1000)
def product(model, x, t, lossf, v):
    params = [p for p in model.parameters() if p.requires_grad]
    out = model(x)
    return curvature.GGNOperator(lossf(out, t), out, params)(v).clone()
m, (x, t), lossf = problem()
names = [n for n, p in m.named_parameters() if p.requires_grad]; sizes = [p.numel() for p in m.parameters() if p.requires_grad]
n = sum(sizes)
v = torch.randn(n, device=dev, generator=torch.Generator(device=dev).manual_seed(7))
t0 = time.time()
m64 = m.double(); ref = product(m64, x.double(), t, lossf, v.double()); torch.cuda.synchronize()
print("fp64 product %.1f s" % (time.time() - t0), flush=True)
scale = float(ref.abs().max())
def report(label, got):
    d = (got.double() - ref).abs()
    worst, off = [], 0
    for nm, sz in zip(names, sizes):
        worst.append((float(d[off:off + sz].max()) / scale, nm)); off += sz
    worst.sort(reverse=True)
    print("RESULT", workload, label, "err %.2e" % (float(d.max()) / scale), " ".join("%s:%.1e" % (k, e) for e, k in worst[:4]), flush=True)
ms, (xs, ts), _ = problem(); report("stock(find)", product(ms, xs, ts, lossf, v))
torch.backends.cudnn.benchmark = False
ms, (xs, ts), _ = problem(); report("stock(immediate)", product(ms, xs, ts, lossf, v))
torch.backends.cudnn.benchmark = True
for label, cl in (("prepared-nchw", False), ("prepared-nhwc", True)):
    pm, (xp, tpp), _ = problem(); modelprep.prepare_model(pm, channels_last=cl); report(label, product(pm, xp, tpp, lossf, v))

"""per-layer accuracy of MIOpen's fp32 convolutions (fwd, bwd-data, wrw) against float64 on the
shapes of a workload, NCHW and NHWC, in find mode"""
import os, sys
sys.path.insert(0, os.getcwd())
import torch
import pytorchhessianfree_amd as hf
from pytorchhessianfree_amd import testproblems as tp
dev = torch.device("cuda", 0)
workload = sys.argv[1]
make = {"resnet18": tp.resnet18_mnist, "allcnnc": tp.allcnnc_cifar100, "resnet50": tp.resnet50_small_images}[workload]
model, (x, t), _ = make(batch_size=32, seed=0, device=dev, data_seed=1000)
shapes = {}
def hook(m, inp, out):
    key = (tuple(inp[0].shape), tuple(m.weight.shape), tuple(m.stride), tuple(m.padding))
    shapes.setdefault(key, 0); shapes[key] += 1
hs = [m.register_forward_hook(hook) for m in model.modules() if isinstance(m, torch.nn.Conv2d)]
with torch.no_grad(): model(x)
for h in hs: h.remove()
g = torch.Generator(device=dev).manual_seed(0)
def rel(a, b): return float((a.double() - b).abs().max() / b.abs().max())
bad = 0
for (xs, ws, st, pd), cnt in shapes.items():
    xx = torch.randn(xs, device=dev, generator=g); ww = torch.randn(ws, device=dev, generator=g) * (2.0 / (ws[1] * ws[2] * ws[3])) ** 0.5
    y64 = torch.nn.functional.conv2d(xx.double(), ww.double(), None, st, pd)
    gy = torch.randn(y64.shape, device=dev, generator=g)
    gx64, gw64, _ = torch.ops.aten.convolution_backward(gy.double(), xx.double(), ww.double(), None, list(st), list(pd), [1, 1], False, [0, 0], 1, [True, True, False])
    for label, fmt in (("nchw", torch.contiguous_format), ("nhwc", torch.channels_last)):
        a, b, c = xx.contiguous(memory_format=fmt), ww.contiguous(memory_format=fmt), gy.contiguous(memory_format=fmt)
        y = torch.nn.functional.conv2d(a, b, None, st, pd)
        gx, gw, _ = torch.ops.aten.convolution_backward(c, a, b, None, list(st), list(pd), [1, 1], False, [0, 0], 1, [True, True, False])
        e = (rel(y, y64), rel(gx, gx64), rel(gw, gw64))
        flag = "BAD" if max(e) > 2e-5 else "ok "
        bad += flag == "BAD"
        if flag == "BAD":
            print("RESULT", flag, label, "x", xs, "w", ws, "s", st, "p", pd, "x%d" % cnt, "fwd %.1e bwd %.1e wrw %.1e" % e, flush=True)
print("RESULT scanned", len(shapes), "shapes, bad entries", bad, flush=True)

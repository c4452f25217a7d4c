"""forward activations of a prepared (NHWC / NCHW) model against the stock model and float64"""
import os, sys
sys.path.insert(0, os.getcwd())
import torch
import pytorchhessianfree_amd as hf
from pytorchhessianfree_amd import modelprep, testproblems as tp
dev = torch.device("cuda", 0)
workload = sys.argv[1]
make = {"resnet18": tp.resnet18_mnist, "allcnnc": tp.allcnnc_cifar100, "resnet50": tp.resnet50_small_images}[workload]
def problem():
    return make(batch_size=32, seed=0, device=dev, data_seed=1000)
def trace(model, x):
    outs = {}
    hooks = []
    for name, m in model.named_modules():
        if len(list(m.children())) == 0:
            hooks.append(m.register_forward_hook(lambda mod, inp, out, name=name: outs.__setitem__(name, out.detach().double().contiguous())))
    with torch.no_grad():
        model(x)
    for h in hooks: h.remove()
    return outs
m64, (x, t), _ = problem(); m64 = m64.double(); ref = trace(m64, x.double())
stock, _, _ = problem(); a = trace(stock, x)
for label, cl in (("nchw", False), ("nhwc", True)):
    pm, _, _ = problem(); modelprep.prepare_model(pm, channels_last=cl); b = trace(pm, x)
    worst = []
    for k in ref:
        if k in a and k in b and ref[k].shape == b[k].shape:
            s = float(ref[k].abs().max()) or 1.0
            worst.append((float((b[k] - ref[k]).abs().max()) / s, float((a[k] - ref[k]).abs().max()) / s, k))
    worst.sort(reverse=True)
    print("RESULT", workload, label, "worst layers (prepared err, stock err):", " ".join("%s:%.1e/%.1e" % (k, e, es) for e, es, k in worst[:6]), flush=True)

import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import pytorchhessianfree_amd as hf
from pytorchhessianfree_amd import modelprep, testproblems as tp
hf.configure()
dev = "cuda"
model, (x, t), lossf = tp.resnet18_mnist(32, device=dev, data_seed=1000)
ref, _, _ = tp.resnet18_mnist(32, device=dev, data_seed=1000)
ref = ref.double()
acts = {}
for name, m in ref.named_modules():
    if isinstance(m, torch.nn.Conv2d):
        m.register_forward_hook(lambda mod, i, o, name=name: acts.__setitem__(name, o.detach()))
ref(x.double())
modelprep.prepare_model(model, channels_last=True, deterministic=(len(sys.argv) > 1))
out = model(x)
for name, m in model.named_modules():
    if isinstance(m, torch.nn.Conv2d):
        y = m._hf_io[1]
        e = float((y.double() - acts[name]).abs().max() / acts[name].abs().max())
        print(f"{name:28s} {tuple(y.shape)} err {e:.2e}")

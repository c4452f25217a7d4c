import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import pytorchhessianfree_amd as hf
from pytorchhessianfree_amd import modelprep, curvature, testproblems as tp
hf.configure()
dev = "cuda"
model, (x, t), lossf = tp.resnet18_mnist(32, device=dev, data_seed=1000)
ref, _, _ = tp.resnet18_mnist(32, device=dev, data_seed=1000)
ref = ref.double()
gref = torch.autograd.grad(lossf(ref(x.double()), t), list(ref.parameters()))
modelprep.prepare_model(model, channels_last=True, deterministic=(len(sys.argv) > 1))
g = torch.autograd.grad(lossf(model(x), t), list(model.parameters()))
sc = max(float(a.abs().max()) for a in gref)
rows = sorted(((float((a.double() - b).abs().max()) / sc, n) for (n, _), a, b in zip(model.named_parameters(), g, gref)), reverse=True)
for r in rows[:8]: print("%.2e %s" % r)

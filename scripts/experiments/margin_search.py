"""Smallest |ReLU input| of the float64 stock model per data seed: data on which two fp32
implementations cannot disagree on a ReLU sign (parity policy)."""
import sys, torch
sys.path.insert(0, ".")
from pytorchhessianfree_amd import testproblems as tp
batch = int(sys.argv[1]); lo = int(sys.argv[2]); hi = int(sys.argv[3])
torch.set_num_threads(4)
model, _, _ = tp.resnet18_mnist(batch, device="cpu"); model = model.double()
best = []
for s in range(lo, hi):
    _, (x, t), _ = None, *[tp.resnet18_mnist(batch, device="cpu", data_seed=s)[1]], None
    m = tp.relu_margin(model, x.double())
    best.append((m, s))
    if m > 5e-7: print(s, "%.2e" % m, flush=True)
best.sort(reverse=True)
print(best[:5])

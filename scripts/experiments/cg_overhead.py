"""Where a Martens-terminated solve's time beyond (iterations x iteration time) goes: cProfile of hf.cg() on the
session's operator (ResNet-18 headline problem), 30 solves."""
import cProfile
import io
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

import pytorchhessianfree_amd as hf  # noqa: E402
from pytorchhessianfree_amd import modelprep  # noqa: E402
from pytorchhessianfree_amd import testproblems as tp  # noqa: E402

model, (x, t), lossf = tp.resnet18_mnist(batch_size=32, device="cuda", data_seed=tp.RESNET18_B32_SEPARATED_SEEDS[0])
modelprep.prepare_model(model, channels_last=True)
opt = hf.HessianFree(model.parameters(), graph_matvec=True)


def forward():
    out = model(x)
    return lossf(out, t), out


op, grad, _loss, sess = opt.linearise(forward)
A = hf.DampedCurvature(op, 1.0)
kw = dict(max_iter=250, martens_conv_crit=True)
for _ in range(3):
    xs, ms, reason = hf.cg(A, -grad, **kw)
n = len(xs) - 1
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(30):
    hf.cg(A, -grad, **kw)
torch.cuda.synchronize()
per = (time.perf_counter() - t0) / 30
print(f"{n} iterations, {reason}: {per * 1e3:.3f} ms per solve = {per / n * 1e6:.1f} us per iteration", flush=True)
pr = cProfile.Profile()
pr.enable()
for _ in range(30):
    hf.cg(A, -grad, **kw)
pr.disable()
torch.cuda.synchronize()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(28)
print(s.getvalue()[:6000])

"""Hessian product of the fused engine on a TRAIN-mode ResNet-18 against float64 double backward of the stock model
(engine's own ReLU decisions replayed), per-parameter error breakdown; timing of the product."""
import os
import sys
import time
import warnings

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import torch  # noqa: E402

from pytorchhessianfree_amd import curvature, modelprep  # noqa: E402
from pytorchhessianfree_amd import testproblems as tp  # noqa: E402
from pytorchhessianfree_amd.engine import FusedGGNEngine  # noqa: E402

DEV = "cuda"
batch = int(sys.argv[1]) if len(sys.argv) > 1 else 32
os.environ.setdefault("HF_ENGINE_DEBUG", "1")
model, (x, t), lossf = tp.resnet18_mnist(batch_size=batch, device=DEV)
model.train()
modelprep.prepare_model(model, channels_last=True)
params = [p for p in model.parameters() if p.requires_grad]
out = model(x)
with warnings.catch_warnings(record=True) as w:
    warnings.simplefilter("always")
    op = curvature.hessian_operator(lossf(out, t), out, params)
    for m in w:
        print("WARN", m.message)
print(type(op).__name__, getattr(op, "hessian", None), getattr(op, "train_bn", None))
v = torch.randn(op.n, device=DEV, generator=torch.Generator(device=DEV).manual_seed(3))
got = op(v).clone()
if isinstance(op, FusedGGNEngine):
    print("repeatable", torch.equal(op(v), got))
    masks = [(u.y > 0) for u in op.units if u.relu]
else:
    masks = None
# float64 reference
from test_engine_gpu import _replay_relu_decisions  # noqa: E402

m64, (x64, t64), lossf64 = tp.resnet18_mnist(batch_size=batch, device=DEV)
m64.train()
m64, x64 = m64.double(), x64.double()
if masks is not None:
    _replay_relu_decisions(m64, masks)
p64 = [p for p in m64.parameters() if p.requires_grad]
o64 = m64(x64)
want = curvature.HessianOperator(lossf64(o64, t64), p64).local(v.double())
err = float((got.double() - want).abs().max() / want.abs().max())
print("max-norm relative error vs float64:", err)
off, worst = 0, []
for name, p in model.named_parameters():
    a, b = got[off:off + p.numel()].double(), want[off:off + p.numel()]
    worst.append((float((a - b).abs().max() / want.abs().max()), name))
    off += p.numel()
print(sorted(worst, reverse=True)[:8])
# stock fp32 autograd for scale
m32, (x32, t32), lossf32 = tp.resnet18_mnist(batch_size=batch, device=DEV)
m32.train()
o32 = m32(x32)
w32 = curvature.HessianOperator(lossf32(o32, t32), [p for p in m32.parameters() if p.requires_grad]).local(v)
print("stock fp32 autograd vs float64:", float((w32.double() - want).abs().max() / want.abs().max()))
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    op(v)
torch.cuda.synchronize()
print("eager product ms:", (time.perf_counter() - t0) / 20 * 1e3)

import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("HF_ENGINE_DEBUG", "1")
import pytorchhessianfree_amd as hf
from pytorchhessianfree_amd import curvature, modelprep, testproblems as tp
hf.configure()
dev = "cuda"
kw = dict(batch_size=4, device=dev, image=32)
model, (x, t), lossf = tp.resnet50_small_images(**kw)
modelprep.prepare_model(model, channels_last=True)
params = [p for p in model.parameters() if p.requires_grad]
out = model(x)
op = curvature.ggn_operator(lossf(out, t), out, params)
print(type(op).__name__)
v = torch.randn(op.n, device=dev, generator=torch.Generator(device=dev).manual_seed(5))
got = op(v).clone()
m64, (x64, t64), _ = tp.resnet50_small_images(**kw)
m64 = m64.double(); p64 = list(m64.parameters()); o64 = m64(x64.double())
want = curvature.GGNOperator(lossf(o64, t64), o64, p64)(v.double())
sc = float(want.abs().max())
names = [n for n, p in model.named_parameters()]
off = 0; rows = []
for n_, p in zip(names, params):
    a, b = got[off:off+p.numel()].double(), want[off:off+p.numel()]
    rows.append((float((a-b).abs().max()) / sc, float(b.abs().max())/sc, n_, tuple(p.shape))); off += p.numel()
for r in sorted(rows, reverse=True)[:12]: print("%.2e (|ref| %.2e) %s %s" % r)
if type(op).__name__ == "FusedGGNEngine":
    for u in op.units[:8] + op.units[-4:]:
        print(u.name, "rows", u.rows, "k", u.cout, "sT sD sW", u.sT, u.sD, u.sW, "rb", u.rb)

"""Time of one graph-replayed ResNet-18 engine product (best of 5 x 200 replays): a quieter A/B signal for
kernel changes than the whole bench."""
import os, sys
sys.path.insert(0, os.getcwd())
import torch
import pytorchhessianfree_amd as hf
from pytorchhessianfree_amd import curvature, modelprep
from pytorchhessianfree_amd import testproblems as tp

hf.configure()
model, (x, t), lossf = tp.resnet18_mnist(batch_size=32, device="cuda", data_seed=tp.RESNET18_B32_SEPARATED_SEEDS[0])
modelprep.prepare_model(model, channels_last=True)
params = [p for p in model.parameters() if p.requires_grad]
def builder():
    out = model(x)
    return curvature.ggn_operator(lossf(out, t), out, params)


op = curvature.GraphedOperator(builder, params=params)
print(type(op).__name__, op.mode[:80])
v = torch.randn(op.n, device="cuda")
res = torch.empty(op.n, device="cuda")
for _ in range(5):
    op.local(v, out=res)
torch.cuda.synchronize()
best = 1e9
for rep in range(5):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(200):
        op.replay_local()
    b.record(); torch.cuda.synchronize()
    best = min(best, a.elapsed_time(b) * 1e3 / 200)
print(f"product: {best:.1f} us (best of 5 x 200 replays)")

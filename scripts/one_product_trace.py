"""Kernels of ONE curvature product of the ResNet-18 workload, in launch order (run under
``rocprofv3 --kernel-trace``; scripts/r2_trace.sh extracts the last product from the trace)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pytorchhessianfree_amd as hf
from pytorchhessianfree_amd import curvature, modelprep, testproblems as tp

hf.configure()
dev = "cuda"
model, (x, t), lossf = tp.resnet18_mnist(32, device=dev, data_seed=tp.RESNET18_B32_SEPARATED_SEEDS[0])
modelprep.prepare_model(model, channels_last=True)
params = [p for p in model.parameters() if p.requires_grad]


def builder():
    out = model(x)
    return curvature.ggn_operator(lossf(out, t), out, params)


op = curvature.maybe_graphed(builder, params=params)
print("operator:", op.mode, file=sys.stderr)
v = torch.randn(op.n, device=dev)
for _ in range(5):
    y = op(v)
torch.cuda.synchronize()
print("checksum", float(y.double().abs().sum()), file=sys.stderr)

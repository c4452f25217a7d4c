"""Time of the engine's v_W scatter (hf_unpack_weights, half 1) alone, graph-replayed; HF_UNPACK_DIRECT=1 selects
the direct gather instead of the LDS-staged walk."""
import os, sys
sys.path.insert(0, os.getcwd())
import torch
import pytorchhessianfree_amd as hf
from pytorchhessianfree_amd import _lib, curvature, modelprep
from pytorchhessianfree_amd import testproblems as tp

hf.configure()
wl = os.environ.get("WORKLOAD", "resnet18")
if wl == "resnet18":
    model, (x, t), lossf = tp.resnet18_mnist(batch_size=32, device="cuda", data_seed=tp.RESNET18_B32_SEPARATED_SEEDS[0])
elif wl == "allcnnc":
    model, (x, t), lossf = tp.allcnnc_cifar100(batch_size=32, device="cuda")
else:
    model, (x, t), lossf = tp.resnet50_small_images(batch_size=32, device="cuda")
modelprep.prepare_model(model, channels_last=True)
params = [p for p in model.parameters() if p.requires_grad]
out = model(x)
op = curvature.ggn_operator(lossf(out, t), out, params)
eng = getattr(op, "engine", None) or getattr(op, "op", op)
v = torch.randn(eng.n, device="cuda")
op.local(v)
for half, slots in ((1, eng._slot_list),):
    for _ in range(3):
        _lib.unpack_tangent(v, slots, half=half)
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        with torch.cuda.graph(g, stream=side):
            for _ in range(10):
                _lib.unpack_tangent(v, slots, half=half)
    torch.cuda.synchronize()
    g.replay(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20):
        g.replay()
    b.record(); torch.cuda.synchronize()
    print(f"{wl} unpack half={half}: {a.elapsed_time(b) * 1e3 / 200:.2f} us ({len(slots)} tensors)")

"""ReLU masks of the prepared fp32 ResNet-18 against the float64 stock model, layer by layer:
a single pre-activation on the other side of zero changes first-order gradients by 1e-4-ish."""
import os, sys, torch
sys.path.insert(0, ".")
import pytorchhessianfree_amd as hf
from pytorchhessianfree_amd import modelprep, testproblems as tp
hf.configure()
dev = "cuda"
seed = int(os.environ.get("DATA_SEED", "1000"))
model, (x, t), lossf = tp.resnet18_mnist(32, device=dev, data_seed=seed)
ref, _, _ = tp.resnet18_mnist(32, device=dev, data_seed=seed); ref = ref.double()
modelprep.prepare_model(model, channels_last=True)
with torch.no_grad():
    model(x)
# fused layers' outputs from the records; reference: recompute by hand
recs = []
bns = [model.bn1] + [b for blk in model.layers for b in (blk.bn1, blk.bn2)]
with torch.no_grad():
    xr = x.double()
    pre = []
    z = ref.bn1(ref.conv1(xr)); pre.append(z); h = ref.maxpool(torch.relu(z))
    for blk in ref.layers:
        idt = h if blk.downsample is None else blk.downsample(h)
        z1 = blk.bn1(blk.conv1(h)); pre.append(z1)
        z2 = blk.bn2(blk.conv2(torch.relu(z1))) + idt; pre.append(z2)
        h = torch.relu(z2)
tot = 0
for i, (bn, z) in enumerate(zip(bns, pre)):
    y = bn._hf_io[2].double()
    flips = int(((y > 0) != (z > 0)).sum())
    err = float((y - torch.relu(z)).abs().max())
    tot += flips
    print(f"layer {i:2d} flips {flips} max abs err {err:.2e} max {float(z.abs().max()):.2f} smallest |z| {float(z.abs().min()):.2e}")
print("total flips", tot)

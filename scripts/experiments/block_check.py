"""Block 0 of the prepared ResNet-18 alone: every intermediate cotangent against float64."""
import os, sys, torch
sys.path.insert(0, ".")
import pytorchhessianfree_amd as hf
from pytorchhessianfree_amd import modelprep, testproblems as tp
hf.configure()
dev = "cuda"
nhwc_in = int(os.environ.get("NHWC_IN", "1"))
model, (x, t), lossf = tp.resnet18_mnist(32, device=dev, data_seed=1000)
ref, _, _ = tp.resnet18_mnist(32, device=dev, data_seed=1000); ref = ref.double()
modelprep.prepare_model(model, channels_last=True)
b0, r0 = model.layers[0], ref.layers[0]
xin0 = torch.randn(32, 64, 7, 7, device=dev)
for rnd in range(2):
    xin = xin0.clone()
    if nhwc_in:
        xin = xin.contiguous(memory_format=torch.channels_last)
    xin.requires_grad_(True)
    xr = xin0.double().requires_grad_(True)
    def pieces(b, x_, fused):
        a1 = b.conv1(x_)
        h1 = modelprep.fused_bn_act(b.bn1, a1, relu=True) if fused else torch.relu(b.bn1(a1))
        a2 = b.conv2(h1)
        y = modelprep.fused_bn_act(b.bn2, a2, res=x_, relu=True, twin=True) if fused else torch.relu(b.bn2(a2) + x_)
        return a1, h1, a2, y
    a1, h1, a2, y = pieces(b0, xin, True)
    ra1, rh1, ra2, ry = pieces(r0, xr, False)
    g = torch.randn(ry.shape, device=dev, dtype=torch.float64, generator=torch.Generator(device=dev).manual_seed(5))
    gs = torch.autograd.grad(y, [a2, h1, a1, xin, b0.conv1.weight, b0.conv2.weight, b0.bn2.bias], g.float().contiguous(memory_format=torch.channels_last))
    gr = torch.autograd.grad(ry, [ra2, rh1, ra1, xr, r0.conv1.weight, r0.conv2.weight, r0.bn2.bias], g)
    print("round", rnd, "fwd", float((y.double() - ry).abs().max() / ry.abs().max()))
    for a, b, n in zip(gs, gr, ["a2", "h1", "a1", "x", "c1.w", "c2.w", "bn2.b"]):
        print("  ", n, "%.2e" % float((a.double() - b).abs().max() / b.abs().max()), tuple(a.stride()))
# whole block through its patched forward
xin = xin0.clone().contiguous(memory_format=torch.channels_last) if nhwc_in else xin0.clone()
xin.requires_grad_(True)
xr = xin0.double().requires_grad_(True)
o, orr = b0(xin), r0(xr)
gs = torch.autograd.grad(o, [xin, b0.conv1.weight, b0.bn2.bias], g.float().contiguous(memory_format=torch.channels_last))
gr = torch.autograd.grad(orr, [xr, r0.conv1.weight, r0.bn2.bias], g)
print("block.forward fwd", float((o.double() - orr).abs().max() / orr.abs().max()))
for a, b, n in zip(gs, gr, ["x", "c1.w", "bn2.b"]):
    print("  ", n, "%.2e" % float((a.double() - b).abs().max() / b.abs().max()))

"""Worker for tests/test_optimizer_gpu.py::test_channels_last_curvature_path_small_net."""

import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

import pytorchhessianfree_amd  # noqa: E402,F401

pytorchhessianfree_amd.configure()
from pytorchhessianfree_amd import _lib, curvature, modelprep  # noqa: E402

DEV = "cuda"


def make():
    torch.manual_seed(0)
    net = torch.nn.Sequential(
        torch.nn.Conv2d(3, 8, 3, padding=1), torch.nn.BatchNorm2d(8), torch.nn.ReLU(),
        torch.nn.Conv2d(8, 16, 3, stride=2, padding=1, bias=False), torch.nn.BatchNorm2d(16),
        torch.nn.ReLU(), torch.nn.Conv2d(16, 6, 1), torch.nn.AdaptiveAvgPool2d(1), torch.nn.Flatten(),
    )
    for m in net.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.running_mean.uniform_(-0.5, 0.5)
            m.running_var.uniform_(0.5, 2.0)
    return net.to(DEV).eval()


stock, fused = make(), make()
modelprep.prepare_model(fused, channels_last=True)
gen = torch.Generator().manual_seed(1)
x = torch.rand(6, 3, 10, 10, generator=gen).to(DEV)
t = torch.randint(0, 6, (6,), generator=gen).to(DEV)
lossf = torch.nn.CrossEntropyLoss()
outs = []
for net in (stock, fused):
    ps = list(net.parameters())
    out = net(x)
    loss = lossf(out, t)
    v = torch.randn(sum(p.numel() for p in ps), generator=torch.Generator().manual_seed(2)).to(DEV)
    grads = curvature.flatten_into(torch.autograd.grad(loss, ps, retain_graph=True), ps)
    outs.append((out.detach(), grads, curvature.GGNOperator(loss, out, ps)(v)))
errors = [float((a - b).abs().max() / b.abs().max()) for a, b in zip(outs[1], outs[0])]

# the gather itself, on a channels_last tensor: exact
w = torch.randn(12, 5, 3, 3, device=DEV).contiguous(memory_format=torch.channels_last)
flat = torch.empty(w.numel() + 7, device=DEV)
_lib.pack(flat, [torch.arange(7.0, device=DEV), w])
exact = bool(torch.equal(flat[7:], w.contiguous().reshape(-1)))

# the NHWC BatchNorm adjoint kernel against the NCHW one on the same numbers
# (vector columns for C % 4 == 0, scalar columns otherwise, with and without the mask)
cl = torch.channels_last
kernel_err = 0.0
for dtype in (torch.float32, torch.float64):
    for (n, c, h) in [(5, 6, 3), (32, 64, 7), (7, 512, 2), (3, 20, 9), (128, 64, 7), (32, 128, 4), (32, 256, 2),
                      (2, 4, 5), (64, 8, 13), (32, 64, 7), (1, 16, 3)]:
        g = torch.Generator(device=DEV).manual_seed(n * c)
        gy, xx, yy = (torch.randn(n, c, h, h, device=DEV, dtype=dtype, generator=g) for _ in range(3))
        mean, w = (torch.randn(c, device=DEV, dtype=dtype, generator=g) for _ in range(2))
        rstd = torch.rand(c, device=DEV, dtype=dtype, generator=g) + 0.5
        for mask in (None, yy):
            ref = modelprep._affine_bwd(gy, xx, mean, rstd, w, mask, need_gres=True)
            got = modelprep._affine_bwd(gy.contiguous(memory_format=cl), xx.contiguous(memory_format=cl), mean, rstd,
                                        w, None if mask is None else mask.contiguous(memory_format=cl), need_gres=True)
            assert got[0].is_contiguous(memory_format=cl) and got[3].is_contiguous(memory_format=cl)
            for a, b in zip(got, ref):
                kernel_err = max(kernel_err, float((a - b).abs().max() / b.abs().max()))
            # two cotangents (residual blocks) and the sums-only mode (conv bias gradients)
            ref2 = modelprep._affine_bwd(gy + xx, xx, mean, rstd, w, mask, need_gres=True)
            got2 = modelprep._affine_bwd(gy.contiguous(memory_format=cl), xx.contiguous(memory_format=cl), mean, rstd,
                                         w, None if mask is None else mask.contiguous(memory_format=cl),
                                         need_gres=True, gy2=xx.contiguous(memory_format=cl))
            for a, b in zip(got2, ref2):
                kernel_err = max(kernel_err, float((a - b).abs().max() / b.abs().max()))
            gb = modelprep._bias_grad(gy.contiguous(memory_format=cl))
            kernel_err = max(kernel_err, float((gb - gy.sum((0, 2, 3))).abs().max() / gy.sum((0, 2, 3)).abs().max()))
# the same kernel captured in a hipGraph and replayed (its scratch is self-resetting)
gyc, xc = (torch.randn(32, 64, 7, 7, device=DEV).contiguous(memory_format=cl) for _ in range(2))
mean, w = torch.randn(64, device=DEV), torch.randn(64, device=DEV)
rstd = torch.rand(64, device=DEV) + 0.5
ref = modelprep._affine_bwd(gyc.contiguous(), xc.contiguous(), mean, rstd, w, None, need_gres=False)
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    modelprep._affine_bwd(gyc, xc, mean, rstd, w, None)  # warm-up: allocates the stream's scratch
side.synchronize()
graph = torch.cuda.CUDAGraph()
with torch.cuda.graph(graph, stream=side):
    out = modelprep._affine_bwd(gyc, xc, mean, rstd, w, None)
for _ in range(4):
    graph.replay()
    torch.cuda.synchronize()
    for a, b in zip(out[:3], ref[:3]):
        kernel_err = max(kernel_err, float((a - b).abs().max() / b.abs().max()))
print("RESULT " + json.dumps({"errors": errors, "gather_exact": exact, "kernel_err": kernel_err}), flush=True)

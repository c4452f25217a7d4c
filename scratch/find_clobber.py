import sys, os
sys.path.insert(0, os.getcwd())
import torch
import pytorchhessianfree_amd
cl = torch.channels_last
shapes = [(64, 64, 3, 1, 1, 7), (64, 128, 3, 2, 1, 7), (128, 128, 3, 1, 1, 4), (64, 128, 1, 2, 0, 7),
          (128, 256, 3, 2, 1, 4), (256, 256, 3, 1, 1, 2), (128, 256, 1, 2, 0, 4), (256, 512, 3, 2, 1, 2), (512, 512, 3, 1, 1, 1), (256, 512, 1, 2, 0, 2)]
B = 32
g = torch.Generator(device="cuda").manual_seed(0)
for (ci, co, k, s, p, H) in shapes:
    xcat = torch.randn(B, 2 * ci, H, H, device="cuda", generator=g).contiguous(memory_format=cl)
    wcat = torch.randn(co, 2 * ci, k, k, device="cuda", generator=g).contiguous(memory_format=cl)
    x0, w0 = xcat.clone(), wcat.clone()
    y = torch.nn.functional.conv2d(xcat, wcat, None, s, p)       # first call: MIOpen find
    torch.cuda.synchronize()
    y2 = torch.nn.functional.conv2d(x0, w0, None, s, p); torch.cuda.synchronize()
    print((ci, co, k, s, p, H), "inputs changed by first call: x", not torch.equal(xcat, x0), "w", not torch.equal(wcat, w0),
          "| first-call output vs second-call output rel %.1e" % float((y - y2).abs().max() / y2.abs().max()))
    gy = torch.randn_like(y)
    xs, ws = xcat[:, :ci].contiguous(memory_format=cl), wcat[:, :ci].contiguous(memory_format=cl)
    xs0, ws0, gy0 = xs.clone(), ws.clone(), gy.clone()
    r = torch.ops.aten.convolution_backward(gy, xs, ws, None, [s, s], [p, p], [1, 1], False, [0, 0], 1, [True, True, False]); torch.cuda.synchronize()
    print("      bwd: inputs changed: gy", not torch.equal(gy, gy0), "x", not torch.equal(xs, xs0), "w", not torch.equal(ws, ws0))

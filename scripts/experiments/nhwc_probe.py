import sys, os, collections, time
sys.path.insert(0, os.getcwd())
import torch
import pytorchhessianfree_amd  # sets winograd off, benchmark on, db
from torch.profiler import profile, ProfilerActivity
cl = int(sys.argv[1])
fmt = torch.channels_last if cl else torch.contiguous_format
# (Cin, Cout, k, stride, pad, H) resnet18 on 28x28: after stem 7x7 maps
shapes = [(1, 64, 7, 2, 3, 28)] + [(64, 64, 3, 1, 1, 7)] * 4 + [(64, 128, 3, 2, 1, 7), (128, 128, 3, 1, 1, 4), (64, 128, 1, 2, 0, 7)] + [(128, 128, 3, 1, 1, 4)] * 2 \
    + [(128, 256, 3, 2, 1, 4), (256, 256, 3, 1, 1, 2), (128, 256, 1, 2, 0, 4)] + [(256, 256, 3, 1, 1, 2)] * 2 \
    + [(256, 512, 3, 2, 1, 2), (512, 512, 3, 1, 1, 1), (256, 512, 1, 2, 0, 2)] + [(512, 512, 3, 1, 1, 1)] * 2
B = 32
items = []
for (ci, co, k, s, p, H) in shapes:
    x = torch.randn(B, ci, H, H, device="cuda").contiguous(memory_format=fmt)
    w = torch.randn(co, ci, k, k, device="cuda").contiguous(memory_format=fmt)
    y = torch.nn.functional.conv2d(x, w, None, s, p)
    gy = torch.randn_like(y).contiguous(memory_format=fmt)
    xc = torch.cat([x, x], 1).contiguous(memory_format=fmt); wc = torch.cat([w, w], 1).contiguous(memory_format=fmt)
    items.append((x, w, gy, xc, wc, s, p))
def run():
    for (x, w, gy, xc, wc, s, p) in items:
        torch.ops.aten.convolution_backward(gy, x, w, None, [s, s], [p, p], [1, 1], False, [0, 0], 1, [True, True, False])
        torch.nn.functional.conv2d(xc, wc, None, s, p)
for _ in range(3): run()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    run(); torch.cuda.synchronize()
c = collections.Counter(); tm = collections.Counter()
for e in prof.events():
    for k in e.kernels: c[k.name[:44]] += 1; tm[k.name[:44]] += k.duration
print("cl", cl, "kernels", sum(c.values()), "device us %.0f" % sum(tm.values()))
for n, k in sorted(c.items(), key=lambda kv: -tm[kv[0]])[:10]: print("   %3d %7.1f %s" % (k, tm[n], n))
g = torch.cuda.CUDAGraph(); s_ = torch.cuda.Stream(); s_.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s_): run()
torch.cuda.synchronize()
with torch.cuda.graph(g, stream=s_): run()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(100): g.replay()
torch.cuda.synchronize(); print("cl", cl, "graph replay ms %.3f" % ((time.perf_counter() - t0) * 10))

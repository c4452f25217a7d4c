import sys, os, time
sys.path.insert(0, os.getcwd())
import torch
import pytorchhessianfree_amd as hf
from pytorchhessianfree_amd import curvature, testproblems as tp, modelprep
m, (x, t), lf = tp.resnet18_mnist(batch_size=32, device="cuda", data_seed=1000)
modelprep.prepare_model(m)
ps = list(m.parameters())
def builder():
    o = m(x); return curvature.GGNOperator(lf(o, t), o, ps)
def timeit(op, n=100):
    for _ in range(5): op.local(op.input_buffer)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): op.local(op.input_buffer)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
one = curvature.GraphedOperator(builder, params=ps)
print("single graph ms", timeit(one))
r1 = one.local(one.input_buffer.normal_()).clone(); v = one.input_buffer.clone()
del one
for frac in (0.75, 0.5, 0.25):
    two = curvature.OverlappedGraphedOperator(builder, params=ps, tail_fraction=frac)
    two.input_buffer.copy_(v)
    r2 = two.local(two.input_buffer)
    print("frac", frac, "cut", two.cut, "tail", (two.n - two.offset) / two.n, "two graphs ms", timeit(two), "rel diff", float((r2 - r1).abs().max() / r1.abs().max()))
    # time phases separately
    for _ in range(3): two.graph.replay()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(50): two.graph.replay()
    torch.cuda.synchronize(); a = (time.perf_counter() - t0) / 50 * 1e3
    t0 = time.perf_counter()
    for _ in range(50): two.graph_head.replay()
    torch.cuda.synchronize(); b = (time.perf_counter() - t0) / 50 * 1e3
    print("   G1 %.3f ms, G2 %.3f ms" % (a, b))
    del two

import sys, os, time, gc
sys.path.insert(0, os.getcwd())
import torch
import pytorchhessianfree_amd as hf
from pytorchhessianfree_amd import curvature, testproblems as tp, modelprep
m, (x, t), lf = tp.resnet18_mnist(batch_size=32, device="cuda")
modelprep.prepare_model(m)
ps = list(m.parameters())
def builder():
    o = m(x); return curvature.GGNOperator(lf(o, t), o, ps)
def T(): torch.cuda.synchronize(); return time.perf_counter()
for rep in range(4):
    t0 = T(); gc.collect(); t1 = T()
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        op = builder(); t2 = T()
        ib = torch.zeros(op.n, device="cuda"); ob = torch.empty(op.n, device="cuda")
        op.local(ib, out=ob); t3 = T()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        op.local(ib, out=ob)
    t4 = T()
    g.replay(); t5 = T()
    print("rep %d: gc %.1f ms, builder %.1f, warm matvec %.1f, capture+instantiate %.1f, first replay %.1f" % (rep, (t1-t0)*1e3, (t2-t1)*1e3, (t3-t2)*1e3, (t4-t3)*1e3, (t5-t4)*1e3))
    del g, op

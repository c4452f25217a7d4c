import sys, os, gc, weakref
sys.path.insert(0, os.getcwd())
import torch
import pytorchhessianfree_amd as hf
from pytorchhessianfree_amd import curvature, testproblems as tp, modelprep
gc.disable()
MB = lambda: torch.cuda.memory_allocated() / 2**20
for prep in (False, True):
    m, (x, t), lf = tp.resnet18_mnist(batch_size=32, device="cuda")
    if prep: modelprep.prepare_model(m)
    ps = list(m.parameters())
    torch.cuda.synchronize(); base = MB()
    o = m(x); loss = lf(o, t)
    op = curvature.GGNOperator(loss, o, ps)
    v = torch.randn(op.n, device="cuda"); r = op(v)
    held = MB()
    del o, loss, op, r, v
    after_del = MB()
    n = gc.collect(); after_gc = MB()
    print("prepared", prep, "MB: base %.0f, with operator %.0f, after del (no gc) %.0f, after gc.collect (%d objs) %.0f" % (base, held, after_del, n, after_gc))

import sys, os, time, warnings
sys.path.insert(0, os.getcwd())
import torch
import pytorchhessianfree_amd as hf
from pytorchhessianfree_amd import testproblems as tp, modelprep
model, (x, t), lossf = tp.resnet18_mnist(batch_size=32, device="cuda")
modelprep.prepare_model(model, channels_last=len(sys.argv) > 1 and sys.argv[1] == "nhwc")
opt = hf.HessianFree(model.parameters(), graph_matvec=True)
g = torch.Generator(device="cuda").manual_seed(0)
t0 = time.time()
for s in range(60):
    xb = torch.rand(32, 1, 28, 28, device="cuda", generator=g); tb = torch.randint(0, 10, (32,), device="cuda", generator=g)
    def forward():
        out = model(xb); return lossf(out, tb), out
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        fl = opt.step(forward)
    if s % 10 == 9:
        print("step %d loss %.4f iters %d alloc %.2f GB reserved %.2f GB  %.1fs" % (s, fl, opt.state["num_cg_iters"][-1], torch.cuda.memory_allocated() / 2**30, torch.cuda.memory_reserved() / 2**30, time.time() - t0), flush=True)

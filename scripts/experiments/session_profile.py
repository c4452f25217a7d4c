"""cProfile + phase timers of session steps (ResNet-18 workload, default settings)."""
import os, sys, time, warnings, cProfile, pstats
sys.path.insert(0, os.getcwd())
import torch
import pytorchhessianfree_amd as hf
from pytorchhessianfree_amd import testproblems as tp, modelprep

hf.configure()
dev = "cuda"
seeds = tp.RESNET18_B32_SEPARATED_SEEDS
model, _, lossf = tp.resnet18_mnist(batch_size=32, device=dev, data_seed=seeds[0])
modelprep.prepare_model(model, channels_last=True)
data = [tp.resnet18_mnist(batch_size=32, device=dev, data_seed=s)[1] for s in seeds]
opt = hf.HessianFree(model.parameters(), graph_matvec=True)
prof = cProfile.Profile()
times = []
for i in range(12):
    x, t = data[i % len(data)]
    def forward():
        o = model(x); return lossf(o, t), o
    torch.cuda.synchronize(); t0 = time.perf_counter()
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        if i >= 4: prof.enable()
        opt.step(forward)
        if i >= 4: prof.disable()
    torch.cuda.synchronize(); times.append(1e3 * (time.perf_counter() - t0))
print("ms", ["%.1f" % a for a in times], "iters", opt.state["num_cg_iters"])
pstats.Stats(prof).sort_stats("cumulative").print_stats(45)

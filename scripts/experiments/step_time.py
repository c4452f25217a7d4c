"""Wall time of complete HessianFree.step() calls on the ResNet-18 workload (default settings: LM
damping, CG-backtracking, line search), per configuration of the curvature product."""
import os, sys, time, warnings
sys.path.insert(0, os.getcwd())
import torch
import pytorchhessianfree_amd as hf
from pytorchhessianfree_amd import testproblems as tp, modelprep, curvature
hf.configure()
import cProfile, pstats
configs = [("stock eager", False, None, "1"), ("prepared NCHW graph", True, False, "1"),
           ("NHWC autograd graph", True, True, "0"), ("NHWC engine graph", True, True, "1")]
which = os.environ.get("ONLY")
for name, graph, cl, engine in configs:
    if which and which not in name:
        continue
    os.environ["HF_ENGINE"] = engine
    model, (x, t), lossf = tp.resnet18_mnist(batch_size=32, device="cuda", data_seed=tp.RESNET18_B32_SEPARATED_SEEDS[0])
    if cl is not None:
        modelprep.prepare_model(model, channels_last=cl)
    def forward():
        out = model(x); return lossf(out, t), out
    opt = hf.HessianFree(model.parameters(), graph_matvec=graph)
    times = []
    prof = cProfile.Profile() if os.environ.get("PROFILE") else None
    for s in range(8):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            if prof and s >= 3: prof.enable()
            fl = opt.step(forward)
            if prof and s >= 3: prof.disable()
        torch.cuda.synchronize(); times.append(time.perf_counter() - t0)
    st = opt.state
    print(f"{name}: step times {['%.3f' % a for a in times]} s; cg iters {st['num_cg_iters']}, losses {st['init_losses'][0]:.4f} -> {fl:.4f}")
    if prof:
        pstats.Stats(prof).sort_stats("cumulative").print_stats(28)

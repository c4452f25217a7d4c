import sys, os, time, warnings
sys.path.insert(0, os.getcwd())
import torch
os.environ.setdefault("MIOPEN_USER_DB_PATH", os.path.join(os.getcwd(), "pytorchhessianfree_amd", "miopen_db"))
torch.backends.cudnn.benchmark = True
import pytorchhessianfree_amd as hf
from pytorchhessianfree_amd import testproblems as tp, modelprep
for graph, fuse in [(False, False), (False, True), (True, True), (True, "nhwc")]:
    model, (x, t), lossf = tp.resnet18_mnist(batch_size=32, device="cuda")
    if fuse: modelprep.prepare_model(model, channels_last=fuse == "nhwc")
    def forward():
        out = model(x); return lossf(out, t), out
    opt = hf.HessianFree(model.parameters(), graph_matvec=graph)
    times = []
    for s in range(6):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            fl = opt.step(forward)
        torch.cuda.synchronize(); times.append(time.perf_counter() - t0)
    st = opt.state
    print(f"graph={graph} fuse_bn={fuse}: step times {['%.3f' % a for a in times]} s; cg iters {st['num_cg_iters']}, reasons {set(st['cg_reasons'])}, losses {['%.4f' % l for l in st['init_losses']]} -> {fl:.4f}, damping {['%.3f' % d for d in st['dampings']]}, best {st['best_cg_iters']}, lr {st['learning_rates']}")

import os, sys, warnings
sys.path.insert(0, os.getcwd())
import torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29517")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
import pytorchhessianfree_amd as hf
from pytorchhessianfree_amd import testproblems as tp, modelprep
for graph in (False, True):
    model, (x, t), lossf = tp.resnet18_mnist(batch_size=16, device="cuda")
    modelprep.prepare_model(model)
    def forward():
        out = model(x); return lossf(out, t), out
    opt = hf.HessianFree(model.parameters(), cg_max_iter=30, process_group=dist.group.WORLD, graph_matvec=graph)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for s in range(3): fl = opt.step(forward)
    print("graph", graph, "ok: iters", opt.state["num_cg_iters"], "loss", opt.state["init_losses"][0], "->", fl, flush=True)
dist.destroy_process_group()

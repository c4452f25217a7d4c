"""Worker for tests/test_optimizer_gpu.py: exercises the RCCL code paths on a 1-rank
group in its OWN process (RCCL's teardown at interpreter exit has been seen to
abort sporadically; a crash there must not take the test session down).
Prints one JSON line."""

import ctypes
import json
import os
import socket
import sys
import warnings

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

import pytorchhessianfree_amd as hf  # noqa: E402

hf.configure()
from pytorchhessianfree_amd import _lib, modelprep  # noqa: E402
from pytorchhessianfree_amd import testproblems as tp  # noqa: E402

DEV = "cuda"
out = {}

# ---- the C ABI's own communicator ------------------------------------------------
lib = _lib.load()
uid = ctypes.create_string_buffer(128)
_lib.check(lib.hf_comm_unique_id(uid), "hf_comm_unique_id")
comm = _lib.c_void_p()
_lib.check(lib.hf_comm_create(ctypes.byref(comm), uid, 1, 0), "hf_comm_create")
v = torch.arange(1000.0, device=DEV)
ref = v.clone()
_lib.check(lib.hf_allreduce_sum(comm, _lib.c_void_p(v.data_ptr()), v.numel(), 0,
                                _lib.current_stream_ptr(v.device)), "hf_allreduce_sum")
torch.cuda.synchronize()
out["abi_allreduce_identity"] = bool(torch.equal(v, ref))
_lib.check(lib.hf_comm_destroy(comm), "hf_comm_destroy")

# ---- HessianFree.step over a torch.distributed (nccl = RCCL) group -------------------
with socket.socket() as s:
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
runs = {}
for name, kw in [("plain", {}), ("dp", dict(process_group=dist.group.WORLD)),
                 ("dp_graph", dict(process_group=dist.group.WORLD, graph_matvec=True))]:
    model, (x, t), lossf = tp.resnet18_mnist(batch_size=8, device=DEV)
    modelprep.prepare_model(model)

    def forward():
        o = model(x)
        return lossf(o, t), o

    opt = hf.HessianFree(model.parameters(), cg_max_iter=15, **kw)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        final = opt.step(forward)
    runs[name] = [opt.state["init_losses"][0], final, opt.state["num_cg_iters"][0],
                  opt.state["cg_reasons"][0]]
out["runs"] = runs

# ---- two-graph overlap with the direct communicators (compute stream + side stream) ---------
from pytorchhessianfree_amd import curvature, distributed as hfdist  # noqa: E402

model, (x, t), lossf = tp.resnet18_mnist(batch_size=8, device=DEV)
modelprep.prepare_model(model)
params = [p for p in model.parameters() if p.requires_grad]


def builder():
    o = model(x)
    return curvature.GGNOperator(lossf(o, t), o, params)


single = curvature.GraphedOperator(builder, params=params)
v = torch.randn(single.n, device=DEV, generator=torch.Generator(device=DEV).manual_seed(3))
want = single(v).clone()
over = curvature.OverlappedGraphedOperator(builder, params=params)
over.group = dist.group.WORLD
got = over(v).clone()
got2 = over(v).clone()
torch.cuda.synchronize()
out["overlap_rel_err"] = float((got - want).abs().max() / want.abs().max())
out["overlap_repeat_rel_err"] = float((got2 - want).abs().max() / want.abs().max())
out["comm_path"] = hfdist.path_name(want, dist.group.WORLD)
out["side_comm"] = hfdist.side_comm(want, dist.group.WORLD) is not None
print("RESULT " + json.dumps(out), flush=True)
dist.destroy_process_group()

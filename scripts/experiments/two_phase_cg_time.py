"""us per PCG iteration of the session's data-parallel product forms inside cg() on a 1-rank RCCL group (the
collective is the identity: what is measured is the hand-over), one two-phase session per pool stream torch hands out
as side stream, with the verdict of the session's own concurrency probe next to it.

    python scripts/experiments/two_phase_cg_time.py            # needs one MI355X
    (profiles/r05_two_phase_side_stream.jsonl)
"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
os.environ.setdefault("RANK", "0")
os.environ.setdefault("WORLD_SIZE", "1")
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

torch.cuda.set_device(0)
dist.init_process_group("nccl")
import pytorchhessianfree_amd as hf  # noqa: E402
from pytorchhessianfree_amd import modelprep  # noqa: E402
from pytorchhessianfree_amd import session as hfsession  # noqa: E402
from pytorchhessianfree_amd import testproblems as tp  # noqa: E402

model, (x, t), lossf = tp.resnet18_mnist(batch_size=32, device="cuda", data_seed=1)
modelprep.prepare_model(model, channels_last=True)


def forward():
    out = model(x)
    return lossf(out, t), out


def run(label, chunk):
    os.environ["HF_CHUNKED_ALLREDUCE"] = chunk
    opt = hf.HessianFree(model.parameters(), graph_matvec=True, process_group=dist.group.WORLD)
    op, grad, _loss, sess = opt.linearise(forward)
    best = 1e9
    for _ in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        hf.cg(hf.DampedCurvature(op, 1e-3), -grad, max_iter=250, martens_conv_crit=False, tol=0.0, atol=0.0)
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / 250 * 1e6)
    print(json.dumps({"variant": label, "us_per_iteration": round(best, 1), "two_phase": sess.split is not None}), flush=True)
    opt._session = None


picked = []
probed_pick = hfsession._concurrent_stream


def next_pool_stream(cur, candidates=8):
    """(experiment) the NEXT pool stream whatever the probe says; the probe's verdict is recorded."""
    cand = torch.cuda.Stream()
    picked.append(bool(hfsession._runs_beside(cand, cur)))
    return cand, picked[-1]  # (round 6: the probe's verdict is returned and kept on the session)


run("single graph + one compact all-reduce", "0")
hfsession._concurrent_stream = next_pool_stream
for k in range(8):
    run(f"two-phase, side stream = pool stream {k}", "1")
    print(json.dumps({"pool_stream": k, "probe_says_concurrent": picked[-1]}), flush=True)
hfsession._concurrent_stream = probed_pick
for k in range(3):
    run(f"two-phase, side stream chosen by the probe (session {k})", "1")
dist.destroy_process_group()

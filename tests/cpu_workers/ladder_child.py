"""Stand-in rank process for the CPU test of ``bench.py``'s data-parallel fallback ladder (``bench.supervise``): joins the
rung's rendezvous exactly as a real rank does (the supervisor's store, this rung's prefix, gloo), then behaves as the
environment says -- ``STUB_FAIL_RUNGS`` / ``STUB_HANG_RUNGS`` / ``STUB_TEARDOWN_CRASH_RUNGS`` (comma-separated rung
numbers; the LAST rank misbehaves) -- and otherwise finishes like a real rank: rank 0 prints one JSON line, every rank
passes the final barrier, writes its marker file and exits."""

import datetime
import json
import os
import sys
import time

import torch
import torch.distributed as dist


def rungs(name):
    return {int(v) for v in os.environ.get(name, "").split(",") if v.strip()}


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    rung = int(os.environ["HF_BENCH_RUNG"])
    last = rank == world - 1
    base = dist.TCPStore(os.environ["MASTER_ADDR"], int(os.environ["MASTER_PORT"]), is_master=False,
                         timeout=datetime.timedelta(seconds=60))
    store = dist.PrefixStore(os.environ["HF_BENCH_STORE_PREFIX"], base)
    dist.init_process_group("gloo", store=store, rank=rank, world_size=world, timeout=datetime.timedelta(seconds=60))
    t = torch.ones(1)
    dist.all_reduce(t)
    assert int(t.item()) == world
    if last and rung in rungs("STUB_FAIL_RUNGS"):
        os._exit(5)
    if last and rung in rungs("STUB_HANG_RUNGS"):
        time.sleep(3600)
    dist.all_reduce(t)  # (a peer of a failed / hanging rank waits here)
    if rank == 0:
        print(json.dumps({"n_gpus": world, "rung": rung, "rungs_failed": json.loads(os.environ["HF_BENCH_RUNGS_FAILED"]),
                          "env": {k: os.environ.get(k) for k in ("HF_CHUNKED_ALLREDUCE", "HF_DIRECT_RCCL", "HF_BENCH_BACKEND")},
                          "argv": sys.argv[1:]}), flush=True)
    dist.barrier()
    with open(os.environ["HF_BENCH_DONE"], "w") as fh:
        fh.write("done\n")
    if last and rung in rungs("STUB_TEARDOWN_CRASH_RUNGS"):
        os.abort()  # (after the marker: the rung counts)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()

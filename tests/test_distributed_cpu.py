"""CPU, 2 processes over gloo: the data-parallel path (batch sharded over ranks,
loss / gradient / every curvature product summed with one all-reduce, identical
PCG on every rank) reproduces the REFERENCE's single-process results on the
concatenated batch -- the reference's own statement of that equivalence is
tests/test_optimizer_acc.py:116-175 ([7, 8] chunks == one batch of 15).

The PCG kernels are GPU-only, so the CPU oracle is plugged into the optimizer's
``_cg`` hook here (as in test_host_logic_cpu.py); on a multi-GPU node the same
code path runs with backend "nccl" (= RCCL) and the HIP kernels (bench.py)."""

import os
import socket
import sys
import warnings

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, mode, curv, reduction, outdir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, HERE)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import pytorchhessianfree_amd as hf
        from conftest import load_golden
        from helpers import T, small_nn, trainable_vec
        from oracle import pcg as oracle

        distinct = mode == "acc_distinct"
        g = load_golden("acc_step_distinct.npz" if distinct else "acc_step.npz")
        key = f"{curv}_{reduction}"
        model = small_nn(g, key)
        lossf = torch.nn.MSELoss(reduction=reduction)
        sizes = [7, 8]
        weight = sizes[rank] / sum(sizes) if reduction == "mean" else 1.0
        opt = hf.HessianFree(model.parameters(), curvature_opt=curv, cg_max_iter=6 if distinct else 4,
                             process_group=dist.group.WORLD, shard_weight=weight)
        opt._cg = oracle.pcg
        out = []
        for s in range(3):
            if distinct:  # rank r holds chunk r of each of the three data lists
                d = {role: [(T(g[f"{key}/{role}_inputs/{s}/{rank}"]), T(g[f"{key}/{role}_targets/{s}/{rank}"]))]
                     for role in ("loss", "grad", "mvp")}
                with warnings.catch_warnings():
                    warnings.simplefilter("ignore")
                    opt.acc_step(model, lossf, d["loss"], grad_datalist=d["grad"], mvp_datalist=d["mvp"],
                                 reduction=reduction)
                out.append(trainable_vec(model).numpy().copy())
                continue
            inputs = T(g[f"{key}/inputs/{s}/{rank}"])
            targets = T(g[f"{key}/targets/{s}/{rank}"])

            def forward():
                o = model(inputs)
                return lossf(o, targets), o

            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                if mode == "step":
                    opt.step(forward)
                else:
                    opt.acc_step(model, lossf, [(inputs, targets)], reduction=reduction)
            out.append(trainable_vec(model).numpy().copy())
        np.savez(os.path.join(outdir, f"rank{rank}.npz"), params=np.stack(out),
                 init_losses=np.array(opt.state["init_losses"]),
                 num_cg_iters=np.array(opt.state["num_cg_iters"]),
                 dampings=np.array(opt.state["dampings"]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["step", "acc_step"])
@pytest.mark.parametrize("curv,reduction", [("ggn", "mean"), ("ggn", "sum"), ("hessian", "mean")])
def test_two_ranks_equal_reference_whole_batch(tmp_path, mode, curv, reduction):
    from conftest import load_golden

    port = _free_port()
    mp.spawn(_worker, args=(2, port, mode, curv, reduction, str(tmp_path)), nprocs=2, join=True)
    r0 = np.load(tmp_path / "rank0.npz")
    r1 = np.load(tmp_path / "rank1.npz")
    # replicas stay bitwise identical (every rank ran the same PCG on the same sums)
    assert np.array_equal(r0["params"], r1["params"])
    g = load_golden("acc_step.npz")
    key = f"{curv}_{reduction}"
    for s in range(3):
        np.testing.assert_allclose(r0["params"][s], g[f"{key}/params_step/{s}"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(r0["init_losses"], g[f"{key}/state_step/init_losses"], rtol=1e-5)
    assert r0["num_cg_iters"].tolist() == g[f"{key}/state_step/num_cg_iters"].tolist()
    np.testing.assert_allclose(r0["dampings"], g[f"{key}/state_step/dampings"], rtol=1e-12)


@pytest.mark.parametrize("curv,reduction", [("ggn", "mean"), ("hessian", "sum")])
def test_two_ranks_acc_step_with_distinct_data_lists(tmp_path, curv, reduction):
    """Every rank passes ITS chunk of the loss / gradient / curvature lists; the result is the
    reference's single-process ``acc_step`` over the whole lists (README.md:147-150 usage)."""
    from conftest import load_golden

    port = _free_port()
    mp.spawn(_worker, args=(2, port, "acc_distinct", curv, reduction, str(tmp_path)), nprocs=2, join=True)
    r0 = np.load(tmp_path / "rank0.npz")
    r1 = np.load(tmp_path / "rank1.npz")
    assert np.array_equal(r0["params"], r1["params"])
    g = load_golden("acc_step_distinct.npz")
    key = f"{curv}_{reduction}"
    for s in range(3):
        np.testing.assert_allclose(r0["params"][s], g[f"{key}/params/{s}"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(r0["init_losses"], g[f"{key}/state/init_losses"], rtol=1e-5)
    assert r0["num_cg_iters"].tolist() == g[f"{key}/state/num_cg_iters"].tolist()
    np.testing.assert_allclose(r0["dampings"], g[f"{key}/state/dampings"], rtol=1e-12)


# ---------------------------------------------------------------------------------------------------------
# The control flow of ``bench.py``'s data-parallel fallback ladder (bench.supervise / bench.RUNGS) with stand-in
# rank processes (tests/cpu_workers/ladder_child.py: the real rendezvous -- the supervisor's store, a prefix per rung,
# gloo -- and scripted failures), both ways in: without a launcher and under ``torch.distributed.run`` (what the
# round driver runs for N > 1: every launched process supervises its own rank, the launcher's store carries the
# rendezvous of every rung and the 'this rung failed' key).  The real thing runs in tests/test_distributed_gpu.py.
# ---------------------------------------------------------------------------------------------------------
def _ladder(env_extra, launcher, rung_timeout="60,60", world=2, timeout=300):
    import json
    import subprocess

    env = {k: v for k, v in os.environ.items()
           if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "TORCHELASTIC_USE_AGENT_STORE")}
    env.update(HF_BENCH_FAKE_DEVICES=str(world), OMP_NUM_THREADS="1",
               HF_BENCH_CHILD_CMD=json.dumps([sys.executable, os.path.join(HERE, "cpu_workers", "ladder_child.py")]),
               **env_extra)
    bench = [os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--steps", "1", "--warmup", "0", "--backend", "gloo",
             "--rung-timeout", rung_timeout]
    if launcher:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
               "--master-addr", "127.0.0.1", "--master-port", str(_free_port())] + bench
    else:
        cmd = [sys.executable] + bench
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout)
    lines = [json.loads(ln) for ln in p.stdout.splitlines() if ln.startswith("{")]
    return p, lines


@pytest.mark.parametrize("launcher", [False, True], ids=["own-launcher", "torchrun"])
def test_bench_ladder_control_flow_with_stand_in_ranks(launcher):
    # nothing fails: the first rung prints the one line
    p, lines = _ladder({}, launcher)
    assert p.returncode == 0 and len(lines) == 1, (p.stdout[-1500:], p.stderr[-3000:])
    assert lines[0]["rung"] == 0 and lines[0]["rungs_failed"] == [] and lines[0]["n_gpus"] == 2
    assert lines[0]["env"] == {"HF_CHUNKED_ALLREDUCE": None, "HF_DIRECT_RCCL": None, "HF_BENCH_BACKEND": "gloo"}
    # the last rank dies on rungs 0 and 1: the third rung (no communicator of the package's own) prints the line and
    # names what failed above it; every supervisor gave the failed rungs up (no orphan waits for the bound)
    p, lines = _ladder({"STUB_FAIL_RUNGS": "0,1"}, launcher)
    assert p.returncode == 0 and len(lines) == 1, (p.stdout[-1500:], p.stderr[-3000:])
    assert lines[0]["rung"] == 2 and [f["rung"] for f in lines[0]["rungs_failed"]] == [0, 1]
    assert lines[0]["env"]["HF_CHUNKED_ALLREDUCE"] == "0" and lines[0]["env"]["HF_DIRECT_RCCL"] == "0"
    # a rank that hangs: the rung's wall-clock bound ends it, the next rung runs
    p, lines = _ladder({"STUB_HANG_RUNGS": "0"}, launcher, rung_timeout="8,60")
    assert p.returncode == 0 and len(lines) == 1, (p.stdout[-1500:], p.stderr[-3000:])
    assert lines[0]["rung"] == 1 and "wall-clock bound" in lines[0]["rungs_failed"][0]["why"]
    # a rank that aborts in its teardown AFTER the run is complete (marker written): the rung counts
    p, lines = _ladder({"STUB_TEARDOWN_CRASH_RUNGS": "0"}, launcher)
    assert p.returncode == 0 and len(lines) == 1 and lines[0]["rung"] == 0, (p.stdout[-1500:], p.stderr[-3000:])
    # every rung fails: non-zero, no line
    p, lines = _ladder({"STUB_FAIL_RUNGS": "0,1,2,3"}, launcher)
    assert p.returncode != 0 and not lines

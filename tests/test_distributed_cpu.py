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

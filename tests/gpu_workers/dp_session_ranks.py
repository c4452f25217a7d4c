"""Worker for tests/test_distributed_gpu.py: ONE rank of a W-rank data-parallel run of
``HessianFree(prepared ResNet-18, graph_matvec=True, process_group=...).step`` -- the drop-in API with
the fused engine, the persistent session and the chunked / overlapped all-reduce behind it.  All ranks
share ``cuda:0`` and talk over gloo (RCCL wants one device per rank); with ``W = 1`` and backend
``nccl`` the same path runs over a real RCCL communicator (grouped launch, side communicator).

The batch of 32 samples of seed ``SEEDS[step]`` is cut into W equal shards (eval-mode BatchNorm: the
curvature is a plain sum over samples, so shards of 16 + 16 equal one batch of 32 -- the statement of
``/root/reference/tests/test_optimizer_acc.py:116-175`` across processes).  Writes ``<outdir>/rank<r>.npz``.

    RANK=r WORLD_SIZE=W MASTER_ADDR=127.0.0.1 MASTER_PORT=p python dp_session_ranks.py <outdir> [mode] [backend]

mode: ``steps`` (default; the two-phase product forced) | ``auto`` (the session's measured choice between the
single-graph and the two-phase product) | ``asym`` (rank 1's session creation is forced to fail on the first step: every
rank must fall back together) | ``acc`` (``acc_step``: every rank passes ITS shard as two chunks -- the
accumulated engine session under data parallelism) | ``frozen`` (stem + layer1 of the model frozen: the engine on a
trainable subset under data parallelism -- single product graph, compact all-reduce of the trainable entries) | ``die``
(the last rank exits mid-run: the others must not hang for good -- used through bench.py's launcher test instead).
"""

import os
import sys
import warnings

HERE = os.path.dirname(os.path.abspath(__file__))
TESTS = os.path.dirname(HERE)
ROOT = os.path.dirname(TESTS)
sys.path.insert(0, ROOT)
sys.path.insert(0, TESTS)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

import pytorchhessianfree_amd as hf  # noqa: E402
from pytorchhessianfree_amd import modelprep  # noqa: E402
from pytorchhessianfree_amd import testproblems as tp  # noqa: E402

hf.configure()
DEV = "cuda:0"
SEEDS = tp.RESNET18_B32_SEPARATED_SEEDS
N_STEPS = 2


def main(outdir, mode="steps", backend="gloo"):
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    if mode != "auto":  # (the two-phase product is what these runs are about; "auto": the measured choice)
        os.environ.setdefault("HF_CHUNKED_ALLREDUCE", "1")
    torch.cuda.set_device(0)
    import datetime

    kw = dict(device_id=torch.device(DEV)) if backend == "nccl" else {}
    dist.init_process_group(backend, rank=rank, world_size=world, timeout=datetime.timedelta(seconds=300), **kw)
    group = dist.group.WORLD
    out = {}
    try:
        model, _, lossf = tp.resnet18_mnist(batch_size=32, device=DEV, data_seed=SEEDS[0])
        if mode == "frozen":
            tp.freeze_stem_and_layer1(model)
        modelprep.prepare_model(model, channels_last=True)
        opt = hf.HessianFree(model.parameters(), graph_matvec=True, process_group=group)
        if mode == "asym" and rank == 1:
            from pytorchhessianfree_amd import session as hfsession

            real = hfsession.EngineSession.try_create
            state = {"n": 0}

            def flaky(*a, **k):
                state["n"] += 1
                return None if state["n"] == 1 else real(*a, **k)

            hfsession.EngineSession.try_create = classmethod(lambda cls, *a, **k: flaky(*a, **k))
        shard = 32 // world
        finals, calls, params, modes = [], [], [], []
        for i in range(N_STEPS):
            _, (x, t), _ = tp.resnet18_mnist(batch_size=32, device=DEV, data_seed=SEEDS[i])
            xs, ts = x[rank * shard:(rank + 1) * shard].contiguous(), t[rank * shard:(rank + 1) * shard].contiguous()

            def forward():
                o = model(xs)
                return lossf(o, ts), o

            sess0 = opt._acc_session if mode == "acc" else opt._session
            c0 = sess0.calls if sess0 is not None else 0
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                if mode == "acc":  # (two chunks per rank: world * 2 chunks accumulate to the whole batch)
                    half = shard // 2
                    chunks = [(xs[:half].contiguous(), ts[:half].contiguous()),
                              (xs[half:].contiguous(), ts[half:].contiguous())]
                    finals.append(opt.acc_step(model, lossf, chunks, reduction="mean"))
                else:
                    finals.append(opt.step(forward))
            sess = opt._acc_session if mode == "acc" else opt._session
            modes.append(0 if sess is None else (2 if getattr(sess, "split", None) is not None else 1))
            calls.append(-1 if sess is None else sess.calls - (c0 if sess is sess0 else 0))
            params.append(torch.cat([p.detach().reshape(-1) for p in opt._params_list]).cpu().numpy().copy())
        st = opt.state
        out["finals"] = np.array(finals)
        out["init_losses"] = np.array(st["init_losses"])
        out["num_cg_iters"] = np.array(st["num_cg_iters"])
        out["best_cg_iters"] = np.array([int(b) for b in st["best_cg_iters"]])
        out["dampings"] = np.array(st["dampings"])
        out["learning_rates"] = np.array(st["learning_rates"])
        out["reasons"] = np.array([str(r) for r in st["cg_reasons"]])
        out["session_mode"] = np.array(modes)      # 0: generic path, 1: session (one graph), 2: session, two-phase
        out["session_calls"] = np.array(calls)
        out["session_off"] = np.array([int(opt._session_off)])
        out["params"] = np.stack(params)
        rep = opt.path_report()["acc_step" if mode == "acc" else "step"] or {}
        out["path"] = np.array([str(rep.get("path")), str((rep.get("data_parallel") or {}).get("product", ""))])
        sess = opt._acc_session if mode == "acc" else opt._session
        if sess is not None:
            from pytorchhessianfree_amd import distributed as hfdist

            out["reduce_bytes"] = np.array([sess.reduce_bytes, 4 * sess.n])
            # the side stream's work should run BESIDE the compute stream's: the verdict of the probe the session ran
            # when it picked the stream (kept on the session; not re-probed here: a wall-clock race)
            if getattr(sess, "side_runs_beside", None) is not None:
                out["side_runs_beside"] = np.array([int(bool(sess.side_runs_beside))])
            val = getattr(sess, "mode_validation", None) or {}
            out["validation"] = np.array([int(bool(val.get("single_graph_identical_on_all_ranks", False))),
                                          int(bool(val.get("two_phase_identical_on_all_ranks", False))),
                                          int(bool(val.get("two_phase_kept_as_candidate", False)))])
            out["validation_rel"] = np.array([float(val.get("two_phase_vs_single_graph", -1.0))])
            timing = getattr(sess, "mode_timing", None)
            out["mode_timing"] = np.array([timing["single_graph_ms"], timing["two_phase_ms"]] if timing else [0.0, 0.0])
            out["comm_path"] = np.array([hfdist.path_name(sess.output_buffer, group)])
            out["side_comm"] = np.array([int(hfdist.side_comm(sess.output_buffer, group) is not None)])
            # the session's product over all ranks against the plain all-reduce of the local products
            v = torch.randn(sess.n, device=DEV, generator=torch.Generator(device=DEV).manual_seed(8))
            plain = sess.local(v).clone()
            dist.all_reduce(plain, group=group)
            got = sess(v).clone()
            out["product_equals_plain_allreduce"] = np.array([bool(torch.equal(got, plain))])
            out["product_rel_err"] = np.array([float((got - plain).abs().max() / plain.abs().max())])
            out["product_checksum"] = np.array([float(got.double().sum()), float(got.double().abs().max())])
        np.savez(os.path.join(outdir, f"rank{rank}.npz"), **out)
        dist.barrier()
    finally:
        dist.destroy_process_group()


if __name__ == "__main__":
    main(*sys.argv[1:4])

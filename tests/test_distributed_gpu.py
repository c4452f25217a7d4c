"""GPU, 2 rank processes on ONE MI355X over gloo: the data-parallel path with the HIP
``cg()`` -- lockstep stop rule, weighted ``hf_pack``, one all-reduce per product, eager
and hipGraph products -- against the REFERENCE's whole-batch traces
(tests/golden/acc_step.npz; the reference states the equivalence in
``/root/reference/tests/test_optimizer_acc.py:116-175``) and rank against rank, bitwise.

Also: ``bench.py --gpus 2`` started WITHOUT a launcher must itself start two ranks and
report ``n_gpus: 2`` (what the round driver runs)."""

import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from tol import within

from conftest import load_golden

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.fixture(scope="module")
def two_rank_run(tmp_path_factory):
    out = tmp_path_factory.mktemp("dp2")
    port = _free_port()
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", LOCAL_RANK=str(r),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen(
            [sys.executable, os.path.join(HERE, "gpu_workers", "dp_two_ranks.py"), str(out)],
            env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    logs = [p.communicate(timeout=600) for p in procs]
    for p, (so, se) in zip(procs, logs):
        assert p.returncode == 0, (so[-2000:], se[-3000:])
    return np.load(out / "rank0.npz"), np.load(out / "rank1.npz")


@pytest.mark.parametrize("mode", ["step", "step_graph", "acc_step"])
@pytest.mark.parametrize("key", ["ggn_mean", "ggn_sum", "hessian_mean"])
def test_two_ranks_hip_pcg_equal_reference_whole_batch(two_rank_run, key, mode):
    r0, r1 = two_rank_run
    g = load_golden("acc_step.npz")
    tag = f"{key}/{mode}/"
    # replicas stay bitwise identical: same sums in, same kernels, same grid
    assert np.array_equal(r0[tag + "params"], r1[tag + "params"])
    assert r0[tag + "num_cg_iters"].tolist() == r1[tag + "num_cg_iters"].tolist()
    ref = f"{key}/state_step/"
    for s in range(3):
        np.testing.assert_allclose(r0[tag + "params"][s], g[f"{key}/params_step/{s}"], rtol=2e-4,
                                   atol=5e-4 * float(np.abs(g[f"{key}/params_step/{s}"]).max()))
    np.testing.assert_allclose(r0[tag + "init_losses"], g[ref + "init_losses"], rtol=1e-5)
    assert r0[tag + "num_cg_iters"].tolist() == g[ref + "num_cg_iters"].tolist()
    assert [str(x) for x in r0[tag + "reasons"]] == [str(x) for x in g[ref + "cg_reasons"]]
    np.testing.assert_allclose(r0[tag + "dampings"], g[ref + "dampings"], rtol=1e-12)
    np.testing.assert_allclose(r0[tag + "learning_rates"], g[ref + "learning_rates"], rtol=1e-12)


def test_lockstep_rule_two_ranks(two_rank_run):
    """A long solve whose operator ends in an all-reduce: both ranks stop at the same
    iteration with the same iterate (bitwise), issue exactly n_iters + LAG + 1 operator
    calls (A(x0), then the lagged stop rule), and agree with the single-process solve of
    the same system."""
    r0, r1 = two_rank_run
    assert np.array_equal(r0["solver/x"], r1["solver/x"])
    assert r0["solver/n_iters"][0] == r1["solver/n_iters"][0]
    assert str(r0["solver/reason"][0]) == str(r1["solver/reason"][0])
    n = int(r0["solver/n_iters"][0])
    assert n > 10
    assert int(r0["solver/calls"][0]) == int(r1["solver/calls"][0]) == n + 2 + 1
    assert np.array_equal(r0["solver/m"], r1["solver/m"])
    # the 2-rank sum rounds differently from the single-process product: fp32-CG close
    assert str(r0["solver/reason"][0]) == str(r0["solver/reason_single"][0])
    within(abs(n - int(r0["solver/n_iters_single"][0])), 3, strict=False)
    x, xs = r0["solver/x"], r0["solver/x_single"]
    within(np.abs(x - xs).max(), 2e-3 * np.abs(xs).max(), strict=False)


def test_engine_product_two_ranks_compact_allreduce(two_rank_run):
    """The fused engine under data parallelism: the structurally-zero entries of the product
    (weight slices of kernel taps that never meet data) stay at home; the result is bitwise the
    plain all-reduce of the two local products, identical on both ranks, and fewer than half of
    the bytes travel on the ResNet-18 workload."""
    r0, r1 = two_rank_run
    assert bool(r0["engine/equal_plain_allreduce"][0]) and bool(r1["engine/equal_plain_allreduce"][0])
    assert np.array_equal(r0["engine/checksum"], r1["engine/checksum"])
    moved, full = (int(x) for x in r0["engine/reduce_bytes"])
    assert moved < 0.5 * full and full == 4 * 11175370


def test_engine_product_two_ranks_chunked_overlapped_allreduce(two_rank_run):
    """The all-reduce chunked by stage (late layers' share first, on its own communicator / asynchronously,
    while the rest of the adjoint sweep runs as a second hipGraph): bitwise the plain all-reduce of the local
    products on 2 ranks, both ranks identical, the overlapped share is most of the bytes, and a 12-iteration
    PCG solve through it (lockstep rule, K1-K3 graph) equals the solve through the single-graph operator."""
    r0, r1 = two_rank_run
    assert bool(r0["chunked/equal_plain_allreduce"][0]) and bool(r1["chunked/equal_plain_allreduce"][0])
    assert float(r0["chunked/tail_share"][0]) > 0.6
    assert bool(r0["chunked/solve_equal"][0]) and bool(r1["chunked/solve_equal"][0])
    assert np.array_equal(r0["chunked/solve_x"], r1["chunked/solve_x"])


def test_bench_gpus_2_starts_two_ranks_itself():
    env = {k: v for k, v in os.environ.items()
           if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run(
        [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1",
         "--iters", "30", "--no-cpu-baseline", "--chunk", "1"],
        env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, (p.stdout[-2000:], p.stderr[-3000:])
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2
    assert rec["config"]["parallelism"].startswith("dp2")
    assert rec["value"] > 0
    # the functional 2-rank run takes the path an 8-GPU run takes: fused engine, chunked compact all-reduce
    # (not the two-graph autograd split a slow gloo collective used to switch on), K1-K3 as one graph
    assert "engine" in rec["config"]["matvec"]
    assert rec["config"]["allreduce"]["overlap_two_graphs"] is False
    assert "product graph A" in rec["config"]["iteration"]
    assert rec["config"]["allreduce"]["bytes"] < 0.5 * 4 * 11175370


# ---------------------------------------------------------------------------------------------------------
# The drop-in API under data parallelism: HessianFree(prepared ResNet-18, graph_matvec=True,
# process_group=...).step -> fused engine + persistent session + chunked / overlapped all-reduce
# ---------------------------------------------------------------------------------------------------------
def _launch_session_ranks(out, world, mode="steps", backend="gloo", timeout=900, env_extra=None, attempts=3):
    """Start ``world`` rank processes of tests/gpu_workers/dp_session_ranks.py and load what they saved.  The ranks are
    started again (up to ``attempts`` times, every restart recorded and counted against the suite's budget:
    tol.note_retry) ONLY for the known teardown abort of RCCL / c10d on this stack: every failed rank was killed by a
    signal and had either written its result file already or names the communication library in its stderr
    (tol.is_teardown_abort).  Any other death -- a rank that exits by itself with an error, a signal inside the
    package's own kernels before the result exists -- fails the test at once."""
    import tol

    for attempt in range(attempts):
        port = _free_port()
        procs = []
        for r in range(world):
            env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK=str(r),
                       MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), **(env_extra or {}))
            procs.append(subprocess.Popen(
                [sys.executable, os.path.join(HERE, "gpu_workers", "dp_session_ranks.py"), str(out), mode, backend],
                env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
        logs = [p.communicate(timeout=timeout) for p in procs]
        codes = [p.returncode for p in procs]
        if all(c == 0 for c in codes):
            return [np.load(out / f"rank{r}.npz") for r in range(world)]
        known = True
        for r, (c, (so, se)) in enumerate(zip(codes, logs)):
            if c != 0:
                print(f"[dp_session_ranks {mode} {backend}] attempt {attempt + 1}: return code {c}\n{se[-4000:]}", flush=True)
                known = known and tol.is_teardown_abort(c, se, (out / f"rank{r}.npz").exists())
        if not known:  # (a real failure)
            break
        tol.note_retry(f"dp_session_ranks {mode} {backend} world {world}", [c for c in codes if c != 0][0], logs[0][1])
        for r in range(world):  # (a fresh attempt must deliver fresh results)
            if (out / f"rank{r}.npz").exists():
                (out / f"rank{r}.npz").unlink()
    for c, (so, se) in zip(codes, logs):
        assert c == 0, (so[-2000:], se[-3000:])


def _check_against_cpu(r0, tol_final=1e-4, family="resnet18"):
    """Against the REAL reference's single-process run on the whole 32-sample batches (tests/golden/
    convnet_resnet18.npz, ``steps``: ``hessianfree.optimizer.HessianFree.step`` x 2 on the stock CPU model).  The
    rank workers build the same model and batches from the same seeds (digests checked in
    tests/test_session_gpu.py::test_session_steps_match_reference_trace)."""
    from helpers import RefTrace

    ref = RefTrace(family, "steps")
    st, finals = ref.state, ref.finals
    n = len(r0["init_losses"])
    np.testing.assert_allclose(r0["init_losses"], st["init_losses"][:n], rtol=1e-5)
    assert [str(x) for x in r0["reasons"]] == [str(x) for x in st["cg_reasons"][:n]]
    assert r0["learning_rates"].tolist() == st["learning_rates"][:n]
    np.testing.assert_allclose(r0["dampings"], st["dampings"][:n], rtol=1e-12)
    for a, b in zip(r0["num_cg_iters"].tolist(), st["num_cg_iters"][:n]):
        within(abs(a - b), 2, strict=False)
    # (the first step 1e-4; a later step starts from fp32-different parameters and back-tracking / the line search
    # pick between nearly tied candidates: 1.6e-4 measured, 5e-4 stated -- only the loss VALUE carries the looser
    # bound: the back-tracked iterate of the first step must be the reference's or its neighbour on the storing grid,
    # the learning rates are compared exactly above)
    for i, (a, b) in enumerate(zip(r0["finals"].tolist(), finals[:n])):
        within(abs(a - b), (tol_final if i == 0 else max(tol_final, 5e-4)) * abs(b), strict=False, note=(i, a, b))
    grid = sorted({int(np.ceil(1.3 ** j)) - 1 for j in range(40)})  # (the storing grid of cg.py:152-170)

    def pos(v):
        return min(range(len(grid)), key=lambda i: abs(grid[i] - int(v)))

    # (the FIRST step only: it starts from the reference's own parameters.  A later step starts from fp32-different
    # ones and walks back over candidates whose losses tie to 1e-4 -- soak lease r6soak9: [12, 14] against [12, 8] on
    # the generic path's MIOpen forwards, final loss within its bound -- so there the loss value is the check)
    for a, b in list(zip(r0["best_cg_iters"].tolist(), st["best_cg_iters"][:n]))[:1]:
        assert abs(pos(a) - pos(b)) <= 1, (r0["best_cg_iters"].tolist(), st["best_cg_iters"][:n])

@pytest.fixture(scope="module")
def session_two_ranks(tmp_path_factory):
    return _launch_session_ranks(tmp_path_factory.mktemp("dps2"), 2)


def test_step_two_ranks_engine_session_equals_cpu_whole_batch(session_two_ranks):
    """Two rank processes, shards of 16 + 16 of each 32-sample batch, two default ``step()`` calls through the
    drop-in API: both ranks take the persistent session in its two-phase mode (chunked / overlapped
    all-reduce), stay bitwise identical, and reproduce the reference's single-process CPU run (golden fixture) on the whole batch
    (stated tolerance: initial losses 1e-5, learning rates / damping schedule / reasons identical, iteration
    counts +-2, final losses 1e-4).  Lockstep: every rank issues exactly n_iters + lag + 1 products per solve
    (lag = 1 for graph-replayed iterations: A(x0), the iterations, one speculative product)."""
    r0, r1 = session_two_ranks
    assert r0["session_mode"].tolist() == [2, 2] and r1["session_mode"].tolist() == [2, 2]
    # (choose_product_mode: both forms identical on both ranks, two-phase equal to the single graph to rounding)
    assert r0["validation"].tolist() == [1, 1, 1] and r1["validation"].tolist() == [1, 1, 1]
    # (HessianFree.path_report(): the path of the step and the data-parallel form the session settled on)
    assert [str(v) for v in r0["path"]] == ["session", "two-phase (chunked / overlapped all-reduce)"]
    assert np.array_equal(r0["params"], r1["params"])
    assert r0["num_cg_iters"].tolist() == r1["num_cg_iters"].tolist()
    for r in (r0, r1):
        assert r["session_calls"].tolist() == [n + 1 + 1 for n in r["num_cg_iters"].tolist()]
        assert bool(r["product_equals_plain_allreduce"][0])
    assert np.array_equal(r0["product_checksum"], r1["product_checksum"])
    moved, full = (int(x) for x in r0["reduce_bytes"])
    assert moved < 0.5 * full
    _check_against_cpu(r0)


def test_step_eight_ranks_on_one_device(tmp_path):
    """world = 8 (BASELINE configs[2]'s world size) as eight rank processes on ONE MI355X over gloo, shards of
    4 samples: the path an 8-GPU run takes -- `phase_split`, two product graphs, the lag rule, compact
    pieces -- functionally.  All ranks bitwise identical; the result equals the whole-batch CPU path within
    the stated tolerances; lockstep call counts on every rank."""
    ranks = _launch_session_ranks(tmp_path, 8, timeout=1500)
    r0 = ranks[0]
    for r in ranks:
        assert r["session_mode"].tolist() == [2, 2]
        assert np.array_equal(r["params"], r0["params"])
        assert r["num_cg_iters"].tolist() == r0["num_cg_iters"].tolist()
        assert r["session_calls"].tolist() == [n + 1 + 1 for n in r["num_cg_iters"].tolist()]
        assert np.array_equal(r["product_checksum"], r0["product_checksum"])
        # (8 addends: the collective's summation order depends on the message layout -- equal to rounding)
        within(float(r["product_rel_err"][0]), 1e-6)
    # (the second step starts from parameters that differ like any two fp32 runs -- here: sums of 8 partial
    # products in the collective's order -- and back-tracking / the line search pick between nearly tied
    # candidates: measured 2.22254 against the CPU path's 2.22289, the LOWER loss; its final loss 5e-4)
    _check_against_cpu(r0, tol_final=5e-4)


def test_step_two_ranks_measured_product_mode(tmp_path):
    """Default policy (``HF_CHUNKED_ALLREDUCE=auto``): at session creation both product forms are timed on the
    group's communicator and the faster is kept -- one decision for all ranks (MAX all-reduce of the timings); the
    steps equal the CPU whole-batch path whichever form won."""
    r0, r1 = _launch_session_ranks(tmp_path, 2, mode="auto")
    assert r0["session_mode"].tolist() == r1["session_mode"].tolist()
    assert set(r0["session_mode"].tolist()) <= {1, 2} and len(set(r0["session_mode"].tolist())) == 1
    assert np.array_equal(r0["mode_timing"], r1["mode_timing"]) and float(r0["mode_timing"].min()) > 0
    two_phase_won = float(r0["mode_timing"][1]) < 0.97 * float(r0["mode_timing"][0])
    assert r0["session_mode"].tolist() == [2 if two_phase_won else 1] * 2
    assert np.array_equal(r0["params"], r1["params"])
    _check_against_cpu(r0)


def test_acc_step_two_ranks_accumulated_engine_session(tmp_path):
    """``acc_step`` under data parallelism on the engine: every rank passes its 16-sample shard as two chunks of 8
    (4 chunks accumulate to the 32-sample batch, weights N_k / sum N over ALL ranks, optimizer.py:677-684), the
    accumulated session serves both calls (one engine per chunk, parallel graph branches, one compact all-reduce
    per product, lockstep), ranks stay bitwise identical and the steps equal the reference's single-process CPU run (golden fixture) on the
    whole batch (same tolerances as ``step``)."""
    r0, r1 = _launch_session_ranks(tmp_path, 2, mode="acc")
    assert r0["session_mode"].tolist() == [1, 1] and r1["session_mode"].tolist() == [1, 1]
    assert np.array_equal(r0["params"], r1["params"])
    assert r0["num_cg_iters"].tolist() == r1["num_cg_iters"].tolist()
    for r in (r0, r1):
        # lockstep through separate launches around the all-reduce: A(x0), the iterations, `lag` speculative ones
        assert all(c >= n + 2 for c, n in zip(r["session_calls"].tolist(), r["num_cg_iters"].tolist()))
        within(float(r["product_rel_err"][0]), 1e-6)
    assert r0["session_calls"].tolist() == r1["session_calls"].tolist()
    assert np.array_equal(r0["product_checksum"], r1["product_checksum"])
    _check_against_cpu(r0, tol_final=5e-4)


def test_step_two_ranks_frozen_layers_engine_session_equals_reference_whole_batch(tmp_path):
    """The engine on a trainable subset under data parallelism: stem + layer1 frozen (N = 11 024 138), shards of
    16 + 16, two default steps through the drop-in API against the REAL reference's whole-batch run on the frozen
    model (golden ``convnet_resnet18_frozen.npz``).  Both ranks take the persistent session in its two-phase mode (the
    late layers' share overlapped with what is left of the adjoint sweep in front of the frozen prefix: layer2), stay
    bitwise identical, only the live trainable entries travel, lockstep call counts."""
    r0, r1 = _launch_session_ranks(tmp_path, 2, mode="frozen")
    assert r0["session_mode"].tolist() == [2, 2] and r1["session_mode"].tolist() == [2, 2]
    assert r0["validation"].tolist() == [1, 1, 1] and bool(r0["product_equals_plain_allreduce"][0])
    assert np.array_equal(r0["params"], r1["params"]) and r0["params"].shape[1] == 11024138
    for r in (r0, r1):
        assert r["session_calls"].tolist() == [n + 1 + 1 for n in r["num_cg_iters"].tolist()]
        within(float(r["product_rel_err"][0]), 1e-6)
    assert np.array_equal(r0["product_checksum"], r1["product_checksum"])
    moved, full = (int(x) for x in r0["reduce_bytes"])
    assert moved < 0.5 * full and full == 4 * 11024138
    _check_against_cpu(r0, family="resnet18_frozen")


def test_step_two_ranks_one_session_refused_falls_back_together(tmp_path):
    """One rank's session creation fails on the first step (the decision must be symmetric): BOTH
    ranks switch the session off for good, take the generic path (engine + compact all-reduce, re-captured per
    step) and still match each other bitwise and the CPU whole-batch path."""
    r0, r1 = _launch_session_ranks(tmp_path, 2, mode="asym")
    assert r0["session_mode"].tolist() == [0, 0] and r1["session_mode"].tolist() == [0, 0]
    assert int(r0["session_off"][0]) == 1 and int(r1["session_off"][0]) == 1
    assert np.array_equal(r0["params"], r1["params"])
    _check_against_cpu(r0)


def test_step_one_rank_rccl_engine_session(tmp_path):
    """The same path over a real RCCL communicator (1-rank ``nccl`` group, channels_last): direct communicator
    on the compute stream, second communicator on the side stream, grouped launch of the tail pieces; the
    session is in its two-phase mode and the steps equal the CPU whole-batch path (a 1-rank shard IS the
    whole batch)."""
    (r0,) = _launch_session_ranks(tmp_path, 1, backend="nccl")
    assert r0["session_mode"].tolist() == [2, 2]
    assert "hf_allreduce_sum" in str(r0["comm_path"][0]) and int(r0["side_comm"][0]) == 1
    # (one pool stream in four shares the compute stream's hardware queue: the session probes for one that does not
    # and keeps the probe's verdict -- a wall-clock race on a GPU that other tests share, so the verdict is recorded,
    # not demanded)
    assert int(r0["side_runs_beside"][0]) in (0, 1)
    # the forms were validated on the live RCCL communicator before the first solve
    assert r0["validation"].tolist() == [1, 1, 1] and 0.0 <= float(r0["validation_rel"][0]) <= 1e-6
    assert bool(r0["product_equals_plain_allreduce"][0])
    assert r0["session_calls"].tolist() == [n + 1 + 1 for n in r0["num_cg_iters"].tolist()]
    _check_against_cpu(r0)


def test_bench_launcher_ends_siblings_when_a_rank_dies():
    """``bench.py --gpus 2`` without a launcher: rank 1 exits non-zero after the process group is up (test hook
    ``HF_BENCH_FAIL_RANK``); the parent must end rank 0 (blocked in its next collective) and return non-zero
    within a bounded time -- fresh child processes only, nothing re-exec'ed."""
    import time

    env = {k: v for k, v in os.environ.items()
           if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["HF_BENCH_FAIL_RANK"] = "1"
    t0 = time.time()
    p = subprocess.run(
        [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1",
         "--iters", "30", "--no-cpu-baseline", "--watchdog", "240"],
        env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode != 0
    assert time.time() - t0 < 300
    assert not [ln for ln in p.stdout.splitlines() if ln.startswith("{")]


# ---------------------------------------------------------------------------------------------------------
# The data-parallel fallback ladder of ``bench.py --gpus N`` (bench.RUNGS): fresh rank processes per rung, a
# wall-clock bound each, the first rung on which every rank reaches the end wins.  Fault injection:
# ``HF_TEST_DP_FAULT`` (session._inject_fault) makes the last rank misbehave in a data-parallel session product.
# ---------------------------------------------------------------------------------------------------------
def _bench_two_ranks(fault, extra=(), timeout=1500):
    env = {k: v for k, v in os.environ.items()
           if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    if fault:
        env["HF_TEST_DP_FAULT"] = fault
    p = subprocess.run(
        [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1",
         "--iters", "20", "--no-cpu-baseline", *extra],
        env=env, capture_output=True, text=True, timeout=timeout)
    assert p.returncode == 0, (p.stdout[-2000:], p.stderr[-4000:])
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["value"] > 0
    ar = rec["config"]["allreduce"]
    assert ar["world"] == 2 and ar["ranks_seen"] == 2
    return rec, ar, p.stderr


def test_bench_ladder_two_phase_mismatch_keeps_single_graph_on_the_first_rung():
    """``mismatch``: the last rank's copy of the two-phase product is off in one entry.  The session's validation on
    the live communicator (choose_product_mode: MIN / MAX all-reduce of a digest) sees that the ranks hold different
    sums, drops the two-phase form with a warning and stays on the single product graph -- same processes, first
    rung, a valid line."""
    rec, ar, err = _bench_two_ranks("mismatch")
    assert ar["rung"] == 0 and ar["rungs_failed"] == []
    v = ar["validation"]
    assert v["single_graph_identical_on_all_ranks"] is True and v["two_phase_identical_on_all_ranks"] is False
    assert v["two_phase_kept_as_candidate"] is False
    assert ar["product_mode"].startswith("single graph")


def test_bench_ladder_falls_to_the_single_graph_rung_when_the_two_phase_form_raises():
    """``raise``: the two-phase product raises on the last rank (first contact: the validation product).  The rank
    process dies, its supervisor gives the rung up for everybody, the second rung (single product graph, fresh
    processes) prints the line and names what failed above it."""
    rec, ar, err = _bench_two_ranks("raise")
    assert ar["rung"] == 1 and [f["rung"] for f in ar["rungs_failed"]] == [0]
    assert "exited with code" in ar["rungs_failed"][0]["why"]
    assert ar["product_mode"].startswith("single graph")
    assert "engine" in rec["config"]["matvec"]


def test_bench_ladder_falls_past_a_rung_that_hangs():
    """``hang``: the last rank never returns from its first two-phase product; its peer waits in the collective.  The
    rung's wall-clock bound ends both, the next rung runs on fresh processes."""
    rec, ar, err = _bench_two_ranks("hang", extra=("--rung-timeout", "75,300"))
    assert ar["rung"] == 1 and [f["rung"] for f in ar["rungs_failed"]] == [0]
    assert "wall-clock bound" in ar["rungs_failed"][0]["why"]


def test_bench_ladder_reaches_the_plainest_rung():
    """``raise:not_plain``: every session product fails on the last rank unless the run is the plainest configuration
    (single graph, torch.distributed.all_reduce, no communicator of the package's own -- ``HF_CHUNKED_ALLREDUCE=0``,
    ``HF_DIRECT_RCCL=0`` in the rank processes' environment): two rungs fail, the third prints the line."""
    rec, ar, err = _bench_two_ranks("raise:not_plain")
    assert ar["rung"] == 2 and [f["rung"] for f in ar["rungs_failed"]] == [0, 1]
    assert "torch.distributed.all_reduce" in ar["path"]


def test_bench_under_torchrun_two_ranks_on_one_device():
    """What the round driver runs for N > 1: ``python -m torch.distributed.run --nproc-per-node N bench.py --gpus N``.
    Every launched process supervises ITS rank as a child (it never touches the GPU), the launcher's store carries the
    rendezvous; rank 0's supervisor prints the one line.  Two ranks share the one device over gloo: functional."""
    env = {k: v for k, v in os.environ.items()
           if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run(
        [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
         "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup",
         "1", "--iters", "20", "--no-cpu-baseline"],
        env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, (p.stdout[-2000:], p.stderr[-4000:])
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    rec = json.loads(lines[0])
    ar = rec["config"]["allreduce"]
    assert rec["n_gpus"] == 2 and ar["rung"] == 0 and ar["rungs_failed"] == [] and ar["ranks_seen"] == 2
    assert ar["validation"]["single_graph_identical_on_all_ranks"] is True
    assert "session" in rec["config"]["path_report"]["step"]["path"]


def test_bench_ladder_rank_process_over_rccl_on_one_gpu():
    """The rank-process code path of a multi-GPU run with the REAL backend, as far as one GPU can take it: ``bench.py
    --gpus 1 --force-dist 1 --ladder 1`` -- a supervisor that never touches the GPU, ONE rank child that joins an
    ``nccl`` (= RCCL) process group through the supervisor's store under the rung's prefix, creates the package's two
    direct communicators, validates both product forms on the live communicator, times them, runs the solves and
    writes its marker.  First rung, validation passed, ``ranks_seen`` = 1, direct RCCL path."""
    env = {k: v for k, v in os.environ.items()
           if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run(
        [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--force-dist", "1", "--ladder", "1", "--steps",
         "1", "--warmup", "1", "--iters", "20", "--no-cpu-baseline", "--no-step-timing", "--no-train-bn"],
        env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, (p.stdout[-2000:], p.stderr[-4000:])
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    rec = json.loads(lines[0])
    ar = rec["config"]["allreduce"]
    assert ar["rung"] == 0 and ar["rungs_failed"] == [] and ar["backend"] == "nccl" and ar["ranks_seen"] == 1
    assert "hf_allreduce_sum" in ar["path"]
    v = ar["validation"]
    assert v["single_graph_identical_on_all_ranks"] and v["two_phase_identical_on_all_ranks"]
    assert v["two_phase_kept_as_candidate"] and v["two_phase_vs_single_graph"] <= 1e-6
    assert ar["product_mode_timing_ms"] and min(ar["product_mode_timing_ms"].values()) > 0

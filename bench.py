"""Benchmark of the hot path: GGN-matvecs/sec (and CG-iters/sec) of the HIP PCG
solver on a ResNet-18-sized parameter vector (BASELINE.json ``configs[1]``).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one Newton-step solve: ``--iters`` (250) PCG iterations on one
synthetic mini-batch, i.e. 251 calls of the damped GGN operator ``A(p)`` (one for
``A(x0)``, cg.py:188) with inputs, weights and all solver vectors resident in
HBM.  ``value`` = damped-operator calls per second summed over ranks; with N>1
every rank holds its own 32-sample shard (weak scaling) and each matvec ends in
one all-reduce of the 44.7 MB partial product.

Rank 0 prints ONE JSON line.  Besides the contract's keys it carries
``roofline`` (dominant hand-written kernel: K2 update_xr, HIP-event timed inside
the timed region) and ``cpu_baseline`` (the oracle -- the reference's algorithm
restated on torch-CPU -- timed on a bounded sample on this box's host cores).
"""

import argparse
import json
import os
import sys
import time
import warnings

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--iters", type=int, default=250, help="PCG iterations per step")
    ap.add_argument("--batch", type=int, default=32, help="samples per GPU")
    ap.add_argument("--workload", default="resnet18", choices=["resnet18", "allcnnc", "resnet50"])
    ap.add_argument("--damping", type=float, default=1e-3,
                    help="Tikhonov damping; 1e-3 keeps all 250 iterations numerically alive "
                         "(with 1.0 this random-init problem converges to fp32 round-off in ~15)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--rung-timeout", default="480,360,300,300",
                    help="multi-rank runs: wall-clock bound in seconds of each rung of the data-parallel fallback ladder "
                         "(comma-separated, the last value repeats; see RUNGS): a rung whose ranks have not all reached "
                         "the end by then is ended and the next one starts with fresh rank processes")
    ap.add_argument("--ladder", type=int, default=-1,
                    help="1: run the rank process(es) under the fallback ladder's supervisor even for one rank (with "
                         "--force-dist 1: the rank-process code path of a multi-GPU run -- RCCL process group on the "
                         "supervisor's store, direct communicators, validation -- on a single GPU); -1: iff N > 1")
    ap.add_argument("--watchdog", type=float, default=900.0,
                    help="multi-rank runs: seconds after which a rank that is still waiting (a peer died, a "
                         "collective hangs) dumps its stack and exits non-zero instead of blocking for good")
    ap.add_argument("--no-step-timing", action="store_true",
                    help="skip the `step_ms` leg (complete default HessianFree.step() calls)")
    ap.add_argument("--cpu-iters", type=int, default=120)
    ap.add_argument("--graph", type=int, default=1, help="replay the matvec as a hipGraph if possible")
    ap.add_argument("--fuse-bn", type=int, default=1,
                    help="modelprep.fuse_eval_batchnorm: eval-mode BN as one fused HIP kernel per pass")
    ap.add_argument("--fuse-conv", type=int, default=1,
                    help="modelprep.fuse_conv_tangent: a conv layer's tangent map as ONE convolution")
    ap.add_argument("--channels-last", type=int, default=-1,
                    help="run the conv layers in NHWC (no MIOpen layout transposes); -1 = where it was "
                         "measured to win and find-db records ship (resnet18 +38 %%, allcnnc +10 %%; resnet50: "
                         "same speed, less accurate NHWC solvers, so NCHW). Any operator is checked against a "
                         "float64 stock-autograd product before it is timed; NHWC falls back to NCHW, and "
                         "config.matvec says which one ran")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL)")
    ap.add_argument("--overlap", type=int, default=-1,
                    help="data parallel: split the product into two hipGraphs and overlap the "
                         "all-reduce of the tail with the head's adjoint sweep (measured on one MI355X: "
                         "the split costs +0.19 ms and the two async collectives +0.4 ms per product, "
                         "so it only pays when the all-reduce itself takes > ~0.6 ms; -1 = decided from a one-off "
                         "timing of the all-reduce at start-up)")
    ap.add_argument("--chunk", type=int, default=-1,
                    help="data parallel + fused engine: 1 = two hipGraphs per product, the all-reduce chunked by stage "
                         "and the late layers' share (ResNet-18: 14 of 17 MB) overlapped with the rest of the adjoint "
                         "sweep on a second communicator; 0 = one product graph + one compact all-reduce; -1 = the "
                         "session times both on this communicator at creation and keeps the faster (what "
                         "HessianFree.step(process_group=...) does by default)")
    ap.add_argument("--force-dist", type=int, default=0,
                    help="create the process group even for WORLD_SIZE=1 (exercises the RCCL path on one GPU)")
    ap.add_argument("--conv", default="", choices=["", "auto", "own", "miopen"],
                    help="implementation of the product's NHWC convolutions: own = the package's one-launch "
                         "deterministic kernels everywhere, miopen = MIOpen everywhere, auto = the faster one "
                         "per layer as measured (default)")
    ap.add_argument("--engine", type=int, default=1,
                    help="1: the fused curvature engine (own deterministic convolutions, BatchNorm fused, "
                         "split-K slabs summed by the consumer kernel) where the model family is supported; "
                         "0: the autograd product")
    ap.add_argument("--no-train-bn", action="store_true",
                    help="skip the `train_bn` leg (the same workload with TRAIN-mode BatchNorm -- what the reference's "
                         "examples/run_resnet18_mnist.py runs -- measured by a child run of this script)")
    ap.add_argument("--bn", default="eval", choices=["eval", "train"],
                    help="BatchNorm mode: eval (running statistics; what batch sharding needs) or train (batch "
                         "statistics, what examples/run_resnet18_mnist.py runs; single GPU)")
    ap.add_argument("--curvature", default="ggn", choices=["ggn", "hessian"],
                    help="curvature_opt of the reference (optimizer.py:25); hessian = BASELINE.json configs[3]")
    ap.add_argument("--precond", type=int, default=0,
                    help="1: diagonal empirical-Fisher preconditioner, exponent 0.75, per-sample autograd "
                         "path (preconditioners.py:63-127), fused into K2/K3 (56 N bytes per iteration)")
    ap.add_argument("--min-calls", type=int, default=1000,
                    help="when a solve ends before --iters (Martens' criterion / round-off on the preconditioned "
                         "Hessian system of config 4), a step repeats the solve back to back until the timed region "
                         "holds at least this many operator calls")
    ap.add_argument("--no-beyond-l3", action="store_true",
                    help="skip the `roofline.frac_beyond_l3` leg (the same PCG kernels on 256 MiB vectors): a rocprofv3 "
                         "--stats run of this command then averages K1 / K2 / K3 over the workload's N only")
    ap.add_argument("--acc", default="",
                    help="comma-separated chunk sizes (e.g. 16,16; they must add up to --batch): drive "
                         "HessianFree.acc_step's path (optimizer.py:519-606) -- loss / gradient / products "
                         "accumulated over the chunks by the accumulated engine session -- instead of step's")
    ap.add_argument("--freeze", default="", choices=["", "stem+layer1"],
                    help="stem+layer1: requires_grad = False on the stem and layer1 of the ResNet workloads -- the "
                         "optimizer then works in the subspace of trainable parameters (optimizer.py:121-123, "
                         "utils.py:31-32); the engine's sweeps start / end at the first trainable layer")
    ap.add_argument("--l2", type=float, default=-1.0,
                    help="L2 regularisation weight added to the loss as in examples/example_utils.py:77-81 "
                         "(-1: 5e-4 for --workload allcnnc with --curvature hessian, else 0)")
    return ap.parse_args()


# The data-parallel fallback ladder of ``bench.py --gpus N`` (N > 1).  The first 8-GPU contact of this path must not be
# able to fail totally: each rung is a FRESH set of rank processes (children of supervisors that never touch the GPU;
# nothing is re-exec'ed) with a wall-clock bound; the first rung on which every rank reaches the end wins, and the line
# says which one it was and what failed above it (``config.allreduce.rung`` / ``rungs_failed``).
RUNGS = [
    # (name, environment of the rank processes, torch.distributed backend or None = --backend)
    ("session picks two-phase (chunked / overlapped all-reduce on a second communicator) or single graph by "
     "validation + timing; direct RCCL on the compute stream", {}, None),
    ("single product graph + one compact all-reduce; direct RCCL on the compute stream",
     {"HF_CHUNKED_ALLREDUCE": "0"}, None),
    ("single product graph + torch.distributed.all_reduce (no communicator of the package's own)",
     {"HF_CHUNKED_ALLREDUCE": "0", "HF_DIRECT_RCCL": "0"}, None),
    ("single product graph + torch.distributed.all_reduce over gloo (host-staged: a functional fallback, not a "
     "scaling number)", {"HF_CHUNKED_ALLREDUCE": "0", "HF_DIRECT_RCCL": "0"}, "gloo"),
]


def supervise(args):
    """Runs the rank processes of a multi-rank bench as CHILDREN, rung by rung (``RUNGS``), and relays rank 0's JSON
    line.  Two ways in: without a launcher (``python bench.py --gpus N``: this one process supervises all N ranks and
    hosts the rendezvous store) and under ``torch.distributed.run`` (every launched process supervises ITS rank; the
    launcher's store carries the rendezvous of every rung under its own prefix, plus a 'this rung failed' key so that
    all supervisors give a rung up together).  A rung counts as done for a rank when its child wrote its marker file
    -- after the final barrier of the run -- whatever the teardown does afterwards.  The supervisor never initialises
    the GPU (``device_count`` only)."""
    import datetime
    import shutil
    import signal
    import subprocess
    import tempfile

    import torch
    from torch.distributed import TCPStore

    launcher = "WORLD_SIZE" in os.environ
    # (test plumbing, tests/test_distributed_cpu.py: the ladder's control flow with stand-in rank processes on a box
    # without a GPU -- HF_BENCH_FAKE_DEVICES / HF_BENCH_CHILD_CMD; the real rank processes are this script itself)
    ndev = int(os.environ.get("HF_BENCH_FAKE_DEVICES", "0")) or torch.cuda.device_count()
    child_cmd = (json.loads(os.environ["HF_BENCH_CHILD_CMD"]) if os.environ.get("HF_BENCH_CHILD_CMD")
                 else [sys.executable, os.path.abspath(__file__)])
    if ndev < 1:
        raise SystemExit("bench.py needs an AMD GPU: the hot path has no CPU fallback")
    if launcher:
        world = int(os.environ["WORLD_SIZE"])
        ranks = [(int(os.environ["RANK"]), int(os.environ.get("LOCAL_RANK", os.environ["RANK"])))]
        addr, port = os.environ.get("MASTER_ADDR", "127.0.0.1"), int(os.environ["MASTER_PORT"])
        store = TCPStore(addr, port, is_master=False, timeout=datetime.timedelta(seconds=300))
    else:
        import socket

        world = args.gpus
        ranks = [(r, r) for r in range(world)]
        with socket.socket() as sock:
            sock.bind(("127.0.0.1", 0))
            port = sock.getsockname()[1]
        addr = "127.0.0.1"
        store = TCPStore(addr, port, is_master=True, wait_for_workers=False,
                         timeout=datetime.timedelta(seconds=300))
    oversubscribed = world > ndev  # several ranks per device: gloo (RCCL wants one device per rank)
    base_backend = "gloo" if (oversubscribed and args.backend == "nccl") else args.backend
    timeouts = [float(v) for v in str(args.rung_timeout).split(",")]
    tmp = tempfile.mkdtemp(prefix="hf_bench_")
    failed, seen_backends = [], set()
    argv = [a for a in sys.argv[1:]]
    for k, (name, env_extra, backend) in enumerate(RUNGS):
        backend = backend or base_backend
        key = (backend, tuple(sorted(env_extra.items())))
        if key in seen_backends:  # (on a shared device every rung already talks gloo: the last rung is the third)
            continue
        seen_backends.add(key)
        if args.chunk >= 0 and k == 0:
            env_extra = dict(env_extra, HF_CHUNKED_ALLREDUCE="1" if args.chunk else "0")
        limit = timeouts[min(k, len(timeouts) - 1)]
        prefix = f"hf_bench/rung{k}"
        procs = []
        for rank, local_rank in ranks:
            marker = os.path.join(tmp, f"rung{k}.rank{rank}.done")
            out_path = os.path.join(tmp, f"rung{k}.rank{rank}.out")
            env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(local_rank), WORLD_SIZE=str(world),
                       MASTER_ADDR=addr, MASTER_PORT=str(port), HF_BENCH_CHILD="1", HF_BENCH_RUNG=str(k),
                       HF_BENCH_RUNG_NAME=name, HF_BENCH_RUNGS_FAILED=json.dumps(failed),
                       HF_BENCH_STORE_PREFIX=prefix, HF_BENCH_DONE=marker, HF_BENCH_BACKEND=backend,
                       HF_BENCH_WATCHDOG=str(limit + 20.0), **env_extra)  # (a last line of defence: the bound is the supervisor's)
            env.pop("TORCHELASTIC_USE_AGENT_STORE", None)  # (the child takes the store it is told, as a client)
            fout = open(out_path, "w")
            procs.append((rank, marker, out_path, fout,
                          subprocess.Popen(child_cmd + argv, env=env, stdout=fout)))
        t0, why = time.time(), None
        fail_key = prefix + "/failed"
        while True:
            states = [(p.poll(), os.path.exists(m)) for _r, m, _o, _f, p in procs]
            if all(done for _rc, done in states) or all(rc is not None for rc, _d in states):
                break
            bad = [(r, p.returncode) for (r, m, _o, _f, p), (rc, done) in zip(procs, states)
                   if rc not in (None, 0) and not done]
            if bad:
                why = f"rank {bad[0][0]} exited with code {bad[0][1]} after {time.time() - t0:.0f} s"
            elif time.time() - t0 > limit:
                why = f"no result within the rung's wall-clock bound of {limit:.0f} s (a rank hangs)"
            if why is not None:
                try:
                    store.set(fail_key, why)
                except Exception:  # noqa: BLE001
                    pass
                break
            try:
                if store.check([fail_key]):
                    why = "another rank's supervisor gave the rung up: " + store.get(fail_key).decode(errors="replace")
                    break
            except Exception:  # noqa: BLE001
                pass
            time.sleep(0.1)
        if why is None and not all(os.path.exists(m) for _r, m, _o, _f, _p in procs):
            bad = [(r, p.returncode) for r, m, _o, _f, p in procs if not os.path.exists(m)]
            why = f"rank {bad[0][0]} ended with code {bad[0][1]} before the end of the run"
            try:
                store.set(fail_key, why)
            except Exception:  # noqa: BLE001
                pass
        # every rank process of this rung ends here, success or not (a finished rank gets a moment for its teardown)
        deadline = time.time() + (20.0 if why is None else 2.0)
        for _r, _m, _o, _f, p in procs:
            while p.poll() is None and time.time() < deadline:
                time.sleep(0.05)
            if p.poll() is None:
                p.send_signal(signal.SIGTERM)
        for _r, _m, _o, fout, p in procs:
            try:
                p.wait(timeout=10)
            except subprocess.TimeoutExpired:
                p.kill()
                p.wait()
            fout.close()
        if why is None:
            for rank, _m, out_path, _f, _p in procs:
                if rank == 0:
                    for ln in open(out_path).read().splitlines():
                        if ln.startswith("{"):
                            print(ln, flush=True)
            shutil.rmtree(tmp, ignore_errors=True)
            raise SystemExit(0)
        print(f"[bench] rung {k} ({name}) failed: {why}", file=sys.stderr, flush=True)
        failed.append({"rung": k, "name": name, "why": why})
    shutil.rmtree(tmp, ignore_errors=True)
    raise SystemExit(3)


def build_problem(args, device, rank):
    from pytorchhessianfree_amd import testproblems as tp

    make = {"resnet18": tp.resnet18_mnist, "allcnnc": tp.allcnnc_cifar100,
            "resnet50": tp.resnet50_small_images}[args.workload]
    # same weights on every rank (seed 0), a different data shard per rank.  The headline workload
    # draws its shards from seeds on which no ReLU input of the float64 model is within fp32
    # rounding of zero (testproblems.relu_margin): the float64 check below compares an fp32 product
    # with a float64 one, and one ReLU sign decided differently moves it by 1e-4 of its max-norm
    seeds = tp.RESNET18_B32_SEPARATED_SEEDS
    if args.workload == "resnet18" and args.batch == 32 and rank < len(seeds):
        prob = make(batch_size=32, seed=0, device=device, data_seed=seeds[rank])
    else:
        prob = make(batch_size=args.batch, seed=0, device=device, data_seed=1000 + rank)
    if args.freeze:
        if args.workload == "allcnnc":
            raise SystemExit("bench --freeze: a ResNet workload")
        tp.freeze_stem_and_layer1(prob[0])
    return prob


def cpu_baseline(args):
    """The oracle (reference algorithm restated, torch CPU ops in the reference's
    order: oracle/pcg.py + oracle/backpack_restated.py) on the same workload,
    bounded to ``--cpu-iters`` PCG iterations."""
    from oracle import backpack_restated as bp
    from oracle import pcg as oracle
    from pytorchhessianfree_amd.utils import vector_to_parameter_list

    from pytorchhessianfree_amd import testproblems as tp

    model, (x, t), lossf = build_problem(args, "cpu", 0)
    if args.bn == "train":
        model.train()
    l2 = args.l2 if args.l2 >= 0 else (5e-4 if (args.workload == "allcnnc" and args.curvature == "hessian") else 0.0)
    if l2 > 0:
        lossf = tp.l2_regularized(lossf, model, l2)
    params = [p for p in model.parameters() if p.requires_grad]
    out = model(x)
    loss = lossf(out, t)
    hessian = args.curvature == "hessian"
    grad = torch.cat([g.reshape(-1) for g in torch.autograd.grad(loss, params, retain_graph=True)])
    calls = [0]

    def mvp(v):  # optimizer.py:450-462 with BackPACK's published algorithms
        calls[0] += 1
        vs = vector_to_parameter_list(v, params)
        if hessian:
            Bv = bp.hessian_vector_product(loss, params, vs)
        else:
            Bv = bp.ggn_vector_product_from_plist(loss, out, params, vs)
        return torch.cat([g.reshape(-1) for g in Bv]).detach()

    M = None
    if args.precond:  # preconditioners.py:63-127, the power re-evaluated on every call as there
        diag = torch.zeros_like(grad)
        for x_i, t_i in zip(x, t):
            g_i = torch.autograd.grad(lossf(model(x_i), t_i), params)
            diag += torch.cat([g.reshape(-1) for g in g_i]) ** 2
        diag /= x.shape[0]

        def M(v):
            return (diag + args.damping) ** -0.75 * v

    lam = args.damping
    # give the CPU path its best thread count on this box (torch's default of one
    # thread per logical core oversubscribes these small convolutions)
    probe = torch.randn_like(grad)
    mvp(probe)  # warm-up
    avail = torch.get_num_threads()
    best = (float("inf"), avail)
    for nt in sorted({c for c in (8, 16, 32, 64, avail) if c <= avail}):
        torch.set_num_threads(nt)
        mvp(probe)
        t0 = time.perf_counter()
        mvp(probe)
        dt = time.perf_counter() - t0
        if dt < best[0]:
            best = (dt, nt)
    torch.set_num_threads(best[1])
    calls[0] = 0
    t0 = time.perf_counter()
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        # same stopping rule as the GPU's timed region: exactly cpu_iters iterations
        xs, _, _ = oracle.pcg(lambda v: mvp(v) + lam * v, -grad, M=M, max_iter=args.cpu_iters, tol=0.0,
                              martens_conv_crit=False, store_x_at_iters=[0])
    dt = time.perf_counter() - t0
    return {
        "value": calls[0] / dt,
        "unit": ("Hessian" if hessian else "GGN") + "-matvecs/s",
        "cores": torch.get_num_threads(),
        "kind": "port",
        "sample": f"{len(xs)-1} PCG iterations ({calls[0]} matvecs) of the same {args.workload} "
                  f"batch-{args.batch} problem, {dt:.1f} s",
        "cg_iters_per_s": (len(xs) - 1) / dt,
        "host_logical_cores": os.cpu_count(),
    }


def beyond_cache_roofline(device, n=64 * 1024 * 1024, iters=48):
    """K2 (and K1, K3) on vectors far beyond the 256 MiB Infinity Cache (six vectors of 256 MiB): the headline's
    six 44.7 MB vectors (268 MB) sit at its edge, so part of the headline fraction is served on-die -- this is
    the plain HBM figure of the same kernels, measured in this run (HIP events inside libhfpcg)."""
    import pytorchhessianfree_amd as hf
    from pytorchhessianfree_amd.cg import enable_kernel_timing, read_kernel_timing

    gen = torch.Generator(device=device).manual_seed(0)
    d = torch.rand(n, device=device, generator=gen) * 100.0 + 1e-3
    b = torch.randn(n, device=device, generator=gen)
    out = torch.empty(n, device=device)
    A = hf.DampedCurvature(lambda v: torch.mul(d, v, out=out), 1e-3)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        hf.cg(A, b, max_iter=8, tol=0.0, store_x_at_iters=[0])
        ws = enable_kernel_timing(device, n, torch.float32, True)
        hf.cg(A, b, max_iter=iters, tol=0.0, store_x_at_iters=[0])
    t = read_kernel_timing(ws)
    enable_kernel_timing(device, n, torch.float32, False)
    res = {"n": n, "vector_MiB": n * 4 / 2**20, "launches_timed": t["n"]}
    for k, by in (("k1", 8), ("k2", 28), ("k3", 12)):
        res[k + "_ms"] = t[k + "_ms"]
        res[k + "_frac"] = by * n / (t[k + "_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS if t[k + "_ms"] > 0 else 0.0
    tot = t["k1_ms"] + t["k2_ms"] + t["k3_ms"]
    res["all_frac"] = 48.0 * n / (tot * 1e-3) / 1e9 / HBM_PEAK_GBS if tot > 0 else 0.0
    del d, b, out
    torch.cuda.empty_cache()
    return res


def full_step_timing(args, device, n_steps=8, warmup=2):
    """Wall time of complete default ``HessianFree.step()`` calls (optimizer.py:126-363: forward,
    gradient, PCG to Martens' criterion, LM damping, CG-backtracking, line search, update) on the
    same workload, a fresh synthetic batch per step -- the persistent engine session where the model
    family is covered.  Reported beside the headline as ``step_ms``."""
    import pytorchhessianfree_amd as hf
    from pytorchhessianfree_amd import modelprep
    from pytorchhessianfree_amd import testproblems as tp

    make = {"resnet18": tp.resnet18_mnist, "allcnnc": tp.allcnnc_cifar100,
            "resnet50": tp.resnet50_small_images}[args.workload]
    seeds = tp.RESNET18_B32_SEPARATED_SEEDS if (args.workload == "resnet18" and args.batch == 32) else range(1000, 1008)
    model, _, lossf = make(batch_size=args.batch, seed=0, device=device, data_seed=seeds[0])
    if args.bn == "train":
        model.train()
    if args.freeze:
        tp.freeze_stem_and_layer1(model)
    l2 = args.l2 if args.l2 >= 0 else (5e-4 if (args.workload == "allcnnc" and args.curvature == "hessian") else 0.0)
    if l2 > 0:
        lossf = tp.l2_regularized(lossf, model, l2)
    modelprep.prepare_model(model, channels_last=bool(args.channels_last))
    batches = [make(batch_size=args.batch, seed=0, device=device, data_seed=sd)[1] for sd in seeds]
    opt = hf.HessianFree(model.parameters(), curvature_opt=args.curvature, graph_matvec=bool(args.graph))
    times = []
    for i in range(warmup + n_steps):
        x, t = batches[i % len(batches)]

        def forward():
            out = model(x)
            return lossf(out, t), out

        acc_sizes = [int(v) for v in args.acc.split(",") if v.strip()]
        chunks, o = [], 0
        for sz in acc_sizes:
            chunks.append((x[o:o + sz].contiguous(), t[o:o + sz].contiguous()))
            o += sz
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            # (config 4: the diagonal empirical-Fisher preconditioner is rebuilt per step at the current
            # damping, per-sample autograd path -- inside the timed step, as a user of the reference pays it)
            M = (opt.get_preconditioner(model, lossf, x, t, "mean", use_backpack=False) if args.precond else None)
            if chunks:
                opt.acc_step(model, lossf, chunks, M_func=M, reduction="mean")
            else:
                opt.step(forward, M_func=M)
        torch.cuda.synchronize()
        times.append((time.perf_counter() - t0) * 1e3)
    timed = times[warmup:]
    st = opt.state
    precond_ms = None
    if args.precond:  # what the preconditioner's construction alone costs, both ways the reference offers
        from pytorchhessianfree_amd import preconditioners

        x, t = batches[0]
        precond_ms = {}
        for name, fn in (("per_sample_autograd (preconditioners.py:63-105; inside every timed step)",
                          preconditioners.diag_EF_autograd),
                         ("batched per-sample gradients (the role of BackPACK's SumGradSquared, preconditioners.py:11-60)",
                          preconditioners.diag_EF_backpack)):
            fn(model, lossf, x, t, "mean")
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            fn(model, lossf, x, t, "mean")
            torch.cuda.synchronize()
            precond_ms[name] = (time.perf_counter() - t0) * 1e3
        if getattr(opt, "_session", None) is not None:
            opt.get_preconditioner(model, lossf, x, t, "mean", use_backpack=False)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            opt.get_preconditioner(model, lossf, x, t, "mean", use_backpack=False)
            torch.cuda.synchronize()
            precond_ms["HessianFree.get_preconditioner with the session's engine (engine.diag_ef: one adjoint sweep + "
                       "per-sample weight-gradient launches; what the timed steps use from the second step on)"] = (
                (time.perf_counter() - t0) * 1e3)
    return {
        "precond_build_ms": precond_ms,
        "mean": sum(timed) / len(timed), "min": min(timed), "max": max(timed), "steps": n_steps,
        "warmup_steps": warmup, "first_step_ms": times[0],
        "cg_iters": st["num_cg_iters"][warmup:], "best_cg_iters": [int(b) for b in st["best_cg_iters"][warmup:]],
        "learning_rates": st["learning_rates"][warmup:],
        "loss_first_to_last": [st["init_losses"][0], st["init_losses"][-1]],
        "mode": ("persistent engine session (engine + graphs kept across steps, graph-replayed forward / "
                 "gradient / trial losses)" if getattr(opt, "_session", None) is not None
                 else "accumulated engine session (acc_step: one engine per chunk, graphs kept across calls)"
                 if getattr(opt, "_acc_session", None) is not None
                 else "generic path (operator rebuilt and re-captured per step, eager trial forwards)"),
        "settings": "HessianFree defaults: damping 1.0 + LM, cg_max_iter 250, Martens' criterion, "
                    "CG-backtracking, line search; a fresh batch per step"
                    + ("; diag empirical-Fisher preconditioner rebuilt per step (per-sample autograd, timed)"
                       if args.precond else ""),
    }


def train_bn_leg(args):
    """The headline workload with TRAIN-mode BatchNorm (the reference's examples/run_resnet18_mnist.py:19-35 never
    calls ``model.eval()``: batch statistics, the GGN couples the samples), measured by a CHILD run of this script
    (a fresh process: its own engine, graphs and MIOpen state; started, never exec'ed) so that the driver's bench run
    times it too.  Same metric, same PCG iteration count; single GPU only (batch statistics do not shard)."""
    import subprocess

    cmd = [sys.executable, os.path.abspath(__file__), "--bn", "train", "--steps", "4", "--warmup", "2",
           "--iters", str(args.iters), "--batch", str(args.batch), "--no-cpu-baseline", "--no-beyond-l3",
           "--no-train-bn"]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    try:
        p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
        lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
        if p.returncode != 0 or not lines:
            return {"error": f"child run failed (rc {p.returncode}): {p.stderr[-400:]}"}
        rec = json.loads(lines[-1])
        step = rec.get("step_ms") or {}
        return {
            "value": rec["value"], "unit": rec["unit"], "ms_per_step": rec["ms_per_step"], "steps": rec["steps"],
            "pcg_iterations_per_step": args.iters, "step_ms": step.get("mean"), "step_ms_cg_iters": step.get("cg_iters"),
            "matvec": rec["config"]["matvec"], "iteration": rec["config"]["iteration"],
            "note": "python bench.py --bn train (child process of this run)",
        }
    except Exception as exc:  # noqa: BLE001  (the headline number must not depend on this leg)
        return {"error": repr(exc)}


def main():
    args = parse()
    if args.conv:
        os.environ["HF_CONV"] = args.conv
    if not args.engine:
        os.environ["HF_ENGINE"] = "0"
    if args.channels_last < 0:
        args.channels_last = 1  # NHWC: the fused engine (ResNets) / +10 % (All-CNN-C), DESIGN.md section 6
    child = os.environ.get("HF_BENCH_CHILD") == "1"
    if not child and (args.gpus > 1 or int(os.environ.get("WORLD_SIZE", "1")) > 1 or args.ladder == 1):
        supervise(args)  # does not return: rank processes are its children, rung by rung
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and rank == 0:
        print(f"[bench] --gpus {args.gpus} but the launcher started {world} ranks; reporting {world}",
              file=sys.stderr, flush=True)
    ndev = torch.cuda.device_count()
    if ndev < 1 or not torch.cuda.is_available():
        raise SystemExit("bench.py needs an AMD GPU: the hot path has no CPU fallback")
    oversubscribed = world > ndev  # several ranks per device: gloo (RCCL wants one device per rank)
    if oversubscribed and args.backend == "nccl":
        args.backend = "gloo"
    if child:
        args.backend = os.environ.get("HF_BENCH_BACKEND", args.backend)
        args.watchdog = min(args.watchdog, float(os.environ.get("HF_BENCH_WATCHDOG", args.watchdog)))
    local_dev = local_rank % ndev
    torch.cuda.set_device(local_dev)
    device = torch.device("cuda", local_dev)
    import pytorchhessianfree_amd as hf

    hf.configure()  # MIOpen: accurate fp32 solvers, find step, shipped find-db (config.py)
    from pytorchhessianfree_amd import curvature, modelprep
    from pytorchhessianfree_amd.cg import _TIMING_SAMPLE, enable_kernel_timing, read_kernel_timing

    from pytorchhessianfree_amd import preconditioners
    from pytorchhessianfree_amd import testproblems as tp

    weight = 1.0 / world
    l2 = args.l2 if args.l2 >= 0 else (5e-4 if (args.workload == "allcnnc" and args.curvature == "hessian") else 0.0)
    hessian = args.curvature == "hessian"
    dist_on = world > 1 or bool(args.force_dist)
    acc_sizes = [int(v) for v in args.acc.split(",") if v.strip()]
    if acc_sizes and sum(acc_sizes) != args.batch:
        raise SystemExit(f"bench --acc {args.acc}: the chunks must add up to --batch {args.batch}")

    def problem(dev, dtype=torch.float32):
        model, (x, t), lossf = build_problem(args, dev, rank)
        model, x = model.to(dtype), x.to(dtype)
        if args.bn == "train":
            model.train()
        if l2 > 0:  # examples/example_utils.py:77-81 (DeepOBS' L2 term on the weights)
            lossf = tp.l2_regularized(lossf, model, l2)
        return model, x, t, lossf

    def make_operator(loss, out, params):
        if hessian:
            return curvature.hessian_operator(loss, out, params, weight=weight, group=None)
        return curvature.ggn_operator(loss, out, params, weight=weight, group=None)

    state = {"group": None, "opt": None}

    def build_operator(channels_last, overlap=False):
        """The operator ``HessianFree.step()`` itself hands to ``cg()`` for this problem -- obtained from
        ``HessianFree.linearise`` (same code path, same process group), not built here: the persistent engine
        session where the model family is covered (single GPU: one product graph cloned into the iteration
        graph; data parallel: two product graphs, the all-reduce chunked by stage and overlapped), else the
        engine / autograd operator as a hipGraph."""
        model, x, t, lossf = problem(device)
        if args.fuse_bn:
            modelprep.fuse_eval_batchnorm(model)
        if args.fuse_conv:
            modelprep.fuse_conv_tangent(model, channels_last=channels_last)
        if args.fuse_bn and args.fuse_conv:
            modelprep.fuse_residual_blocks(model)  # relu(bn(.)) / relu(bn(.) + identity) as one layer
            modelprep.fuse_bn_relu(model)          # relu(bn(.)) outside residual blocks (the stem)
            modelprep.skip_identity_pools(model)   # AdaptiveAvgPool2d(1) of a 1x1 map
            modelprep._install_engine_hooks(model)  # lets the optimizer use the fused engine / session
        params = [p for p in model.parameters() if p.requires_grad]
        group = state["group"]
        diag = None
        if args.precond:  # per-sample autograd path, preconditioners.py:63-105
            diag = preconditioners.diag_EF_autograd(model, lossf, x, t, "mean") * weight
            if group is not None:
                torch.distributed.all_reduce(diag, group=group)

        def forward():
            out = model(x)
            return lossf(out, t), out

        if args.graph and overlap and not hessian:
            # EXPERIMENTAL (--overlap 1 only): the two-graph split of the AUTOGRAD sweeps, built here
            os.environ["HF_ENGINE"] = "0"

            def builder():
                out = model(x)
                return curvature.ggn_operator(lossf(out, t), out, params, weight=weight, group=group)

            grad = curvature.flatten_into(torch.autograd.grad(lossf(model(x), t), params), params, scale=weight)
            if group is not None:
                torch.distributed.all_reduce(grad, group=group)
            op = curvature.OverlappedGraphedOperator(builder, params=params)
            return op, grad, diag, sum(p.numel() for p in params)
        opt = hf.HessianFree(model.parameters(), curvature_opt=args.curvature, graph_matvec=bool(args.graph),
                             process_group=group, shard_weight=weight if group is not None else None)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            if acc_sizes:
                chunks, o = [], 0
                for sz in acc_sizes:
                    chunks.append((x[o:o + sz].contiguous(), t[o:o + sz].contiguous()))
                    o += sz
                _fwd, grad, op, sess = opt.acc_linearise(model, lossf, chunks, reduction="mean")
                if sess is None:
                    raise SystemExit("bench --acc: the accumulated engine session does not cover this workload")
                op, grad = sess, sess.gradient()
            else:
                op, grad, _loss, _sess = opt.linearise(forward)
        state["opt"], state["model"] = opt, model  # (the session lives as long as its optimizer)
        return op, grad, diag, sum(p.numel() for p in params)

    def stock_product(v, dtype, masks=None, record=None):
        """The same product by stock PyTorch-ROCm autograd on an unpatched NCHW model
        (float64: the reference; float32: what the stock fp32 path itself achieves).

        ``masks`` / ``record``: the sign decisions of the model's ReLU calls, in call order -- replayed
        from a list / appended to one.  Two correct fp32 forward passes may decide a ReLU whose input lies
        within rounding of zero differently, and ONE such decision moves a curvature product of a deep net
        by ~1e-4 of its max-norm (it is why products of the 50-layer net scatter between 1e-6 and 3e-4 of
        the float64 product from run to run, for every implementation): an operator is therefore compared
        with the float64 product that takes ITS OWN ReLU decisions -- the same piecewise-linear network."""
        import types

        model, x, t, lossf = problem(device, dtype)
        if masks is not None or record is not None:
            cursor = [0]

            def relu_forward(self, inp):
                if masks is not None:
                    m = masks[cursor[0]]
                    cursor[0] += 1
                else:
                    m = inp > 0
                    record.append(m)
                return inp * m.to(inp.dtype)

            for mod in model.modules():
                if isinstance(mod, torch.nn.ReLU):
                    mod.forward = types.MethodType(relu_forward, mod)
        params = [p for p in model.parameters() if p.requires_grad]
        find = torch.backends.cudnn.benchmark
        # one product only: MIOpen's immediate mode, no find step (and no find-db records)
        # for the shapes of PyTorch's generic double-backward
        torch.backends.cudnn.benchmark = False
        try:
            out = model(x)
            op = make_operator(lossf(out, t), out, params)
            return op(v.to(dtype)).double().clone()
        finally:
            torch.backends.cudnn.benchmark = find

    def relu_decisions(op):
        """The ReLU sign decisions of a fused-engine operator, in the model's call order (``None`` for
        operators that do not expose them: the plain float64 product is the reference then)."""
        engines = getattr(op, "engines", None)
        if engines:  # (accumulated session: one engine per chunk, the chunks in batch order)
            return [torch.cat([(e.units[i].y > 0) for e in engines]) for i, u in enumerate(engines[0].units) if u.relu]
        eng = getattr(op, "engine", None) or getattr(op, "op", op)
        units = getattr(eng, "units", None)
        if not units or "engine" not in getattr(eng, "mode", ""):
            return None
        return [(u.y > 0) for u in units if u.relu]

    # guard: an operator is only timed after it reproduced a float64 stock-autograd product
    # as well as stock fp32 autograd does (x5, floor 1e-5; deep random-init nets such as the
    # ResNet-50 workload are only good to ~1e-4 in fp32, ResNet-18 to 2e-7).  A wrong operator
    # must never be what gets measured; NHWC falls back to NCHW, NCHW aborts.
    check = {}

    def reference_products():
        """float64 and fp32 stock-autograd products of one random vector: the tolerance of every
        operator check in this run is max(1e-5, 5 x what stock fp32 autograd itself achieves) -- the
        engine's own first-use guard included."""
        model, _, _, _ = problem(device)
        n = sum(p.numel() for p in model.parameters() if p.requires_grad)
        del model
        v = torch.randn(n, device=device, generator=torch.Generator(device=device).manual_seed(7))
        want = stock_product(v, torch.float64)
        scale = float(want.abs().max())
        rec = []
        stock32 = stock_product(v, torch.float32, record=rec)
        # stock fp32 autograd against the float64 product on ITS ReLU decisions: the fp32 noise floor of
        # the workload; `stock_err_plain` (against the float64 network's own decisions) for the record
        stock_err = float((stock32 - stock_product(v, torch.float64, masks=rec)).abs().max()) / scale
        check.update(v=v, want=want, scale=scale, stock_err=stock_err, tol=max(1e-5, 5.0 * stock_err),
                     stock_err_plain=float((stock32 - want).abs().max()) / scale)
        del rec, stock32
        from pytorchhessianfree_amd.engine import FusedGGNEngine

        FusedGGNEngine.verify_tol = check["tol"]

    def checked(channels_last, overlap=False):
        if "want" not in check:
            reference_products()
        op, grad, diag, n = build_operator(channels_last, overlap)
        # (this rank's own weighted product: the float64 reference below is of this rank's shard)
        got = op.local(check["v"]).double().clone()
        masks = relu_decisions(op)
        ref = check["want"] if masks is None else stock_product(check["v"], torch.float64, masks=masks)
        err = float((got - ref).abs().max()) / check["scale"]
        check["err_plain"] = float((got - check["want"]).abs().max()) / check["scale"]
        check["masked"] = masks is not None
        # (an operator that does not expose its ReLU decisions is held to what stock fp32 autograd achieves
        # against the same plain float64 product)
        check["tol_eff"] = check["tol"] if masks is not None else max(1e-5, 5.0 * check["stock_err_plain"])
        del masks, ref
        # the reference's own check (optimizer.py:414-448): the same product twice
        check["deterministic"] = bool(torch.equal(op.local(check["v"]).double(), got))
        return op, grad, diag, n, err

    group, allreduce_ms, comm_path, ranks_seen = None, None, "none", 1
    if dist_on:
        import torch.distributed as dist

        from pytorchhessianfree_amd import distributed as hfdist

        if world == 1:  # --force-dist on a single GPU, started without a launcher
            import socket

            with socket.socket() as sock:
                sock.bind(("127.0.0.1", 0))
                free_port = sock.getsockname()[1]
            for key, val in (("RANK", "0"), ("WORLD_SIZE", "1"), ("MASTER_ADDR", "127.0.0.1"),
                             ("MASTER_PORT", str(free_port))):
                os.environ.setdefault(key, val)
        # every wait of this process is bounded: collectives issued through torch.distributed by the
        # group's timeout, everything else (direct RCCL calls drained by a stream synchronisation) by a
        # wall-clock watchdog that dumps the stacks and ends the process with a non-zero code -- the
        # launcher above then ends the sibling ranks; nothing is ever re-exec'ed
        import datetime
        import faulthandler

        faulthandler.dump_traceback_later(args.watchdog, exit=True)
        limit = datetime.timedelta(seconds=args.watchdog)
        # (gloo and RCCL print connection banners on fd 1: stdout carries the ONE JSON line only)
        sys.stdout.flush()
        saved_fd = os.dup(1)
        os.dup2(2, 1)
        try:
            kw = {}
            if child:  # (a rung of the ladder: the supervisor's store, this rung's keys under their own prefix)
                base = dist.TCPStore(os.environ["MASTER_ADDR"], int(os.environ["MASTER_PORT"]), is_master=False,
                                     timeout=limit)
                kw = dict(store=dist.PrefixStore(os.environ["HF_BENCH_STORE_PREFIX"], base), rank=rank, world_size=world)
            if args.backend == "nccl":
                dist.init_process_group("nccl", device_id=device, timeout=limit, **kw)
            else:
                dist.init_process_group(args.backend, timeout=limit, **kw)
            dist.barrier()
        finally:
            sys.stdout.flush()
            os.dup2(saved_fd, 1)
            os.close(saved_fd)
        group = dist.group.WORLD
        world = dist.get_world_size(group)  # what the backend reports is what gets printed
        state["group"] = group
        if os.environ.get("HF_BENCH_FAIL_RANK") == str(rank):  # test hook: a rank that dies mid-run (on every rung)
            os._exit(3)
        # the data-parallel product: --chunk 1 two-phase (chunked / overlapped all-reduce), 0 single graph + one
        # compact all-reduce, -1 (default) whichever the session validates and measures to be faster on this
        # communicator (a rung of the ladder below the first has it in its environment already)
        if args.chunk >= 0 and not (child and int(os.environ.get("HF_BENCH_RUNG", "0")) > 0):
            os.environ["HF_CHUNKED_ALLREDUCE"] = "1" if args.chunk else "0"
    args.overlap = max(args.overlap, 0)  # (the autograd two-graph split: only on request)
    op, grad, diag, n, err = checked(bool(args.channels_last), overlap=bool(args.overlap))
    def note_text(err):
        if check.get("masked"):
            return ("max-norm error against float64 stock autograd on this operator's own ReLU decisions {:.1e} "
                    "(on the float64 network's decisions {:.1e}); stock fp32 autograd likewise {:.1e} ({:.1e})"
                    ).format(err, check["err_plain"], check["stock_err"], check["stock_err_plain"])
        return "max-norm error against float64 stock autograd {:.1e}, stock fp32 autograd {:.1e}".format(
            err, check["stock_err_plain"])

    class _Note:  # (keeps the two call sites below unchanged)
        @staticmethod
        def format(err, _unused):
            return note_text(err)

    note = _Note
    layout = ("NHWC" if args.channels_last else "NCHW") + " (" + note.format(err, check["stock_err"]) + ")"
    if not err < check["tol_eff"] and args.channels_last:
        print(f"[bench] NHWC product off by {err:.2e} (float64 reference); using NCHW",
              file=sys.stderr, flush=True)
        del op
        args.channels_last = 0
        op, grad, diag, n, err = checked(False)
        layout = "NCHW (NHWC failed its check; " + note.format(err, check["stock_err"]) + ")"
    if not err < check["tol_eff"]:
        raise SystemExit(f"bench: the curvature product is off by {err:.2e} against float64 stock autograd")
    del check["want"], check["v"]

    if group is not None:
        # one-off timing of THE collective of this path: the operator's own reduction of a product
        # (all-reduce of the 4N-byte vector; the engine moves only the entries that can be non-zero)
        probe = torch.zeros(n, device=device)
        for _ in range(3):
            op.reduce(probe)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            op.reduce(probe)
        torch.cuda.synchronize()
        tt = torch.tensor([(time.perf_counter() - t0) / 10 * 1e3], dtype=torch.float64, device=device)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX, group=group)  # one number, on every rank
        allreduce_ms = float(tt.item())
        comm_path = hfdist.path_name(probe, group)
        ranks_seen = hfdist.ranks_seen(group, device)
        del probe
        if ranks_seen != world:  # (before anything is printed: such a run is not a data-parallel run)
            raise SystemExit(f"bench: world {world} but the product's collective sums over {ranks_seen} rank(s)")
    b = -grad
    A = hf.DampedCurvature(op, args.damping)
    M = hf.DiagonalPreconditioner(diag, args.damping, 0.75) if diag is not None else None

    def solve(martens=False, max_iter=args.iters):
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            return hf.cg(A, b, M=M, max_iter=max_iter, tol=0.0, martens_conv_crit=martens,
                         store_x_at_iters=[0])

    def barrier():
        if group is not None:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    calls_w = op.calls
    for _ in range(args.warmup):
        xs_w, _, reason_w = solve()
    # A solve that cannot sustain --iters iterations (config 4: fp32 round-off ends the preconditioned Hessian
    # solve at tol = 0 in "Divergence" after ~21 iterations) is replaced by what the optimizer actually runs --
    # Martens-terminated solves -- repeated back to back so that the timed region holds >= --min-calls calls
    sustained = args.warmup == 0 or len(xs_w) - 1 >= args.iters
    timed_martens = not sustained
    solves_per_step = 1
    if not sustained:
        c0 = op.calls
        solve(martens=True)
        per = max(1, op.calls - c0)
        solves_per_step = max(1, -(-args.min_calls // (per * max(1, args.steps))))
    ws = enable_kernel_timing(device, n, torch.float32, not os.environ.get("HF_BENCH_NO_TIMING"))
    calls0 = op.calls
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        for _r in range(solves_per_step):
            xs, _, reason = solve(martens=timed_martens)
    barrier()
    dt = time.perf_counter() - t0
    timing = read_kernel_timing(ws)
    enable_kernel_timing(device, n, torch.float32, False)
    matvecs = op.calls - calls0
    iters_done = len(xs) - 1

    if group is not None:
        tt = torch.tensor([dt], device=device, dtype=torch.float64)
        torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
        dt = float(tt.item())

    # CG-iters/sec to Martens' criterion (second half of BASELINE.json's metric), twice: at the timed region's damping
    # (--damping; the reference's fixture `solve_martens` of this problem: 36 iterations at 1e-3) and at the
    # optimizer's DEFAULT damping 1.0 (the reference's own trace of this problem never reaches Martens'
    # test there: it ends by tolerance after 12-15 iterations -- cg.py:95-115 evaluates Martens' test first, then the
    # iteration count, NaN, the tolerances), through exactly the call `HessianFree.step` makes (tol 1e-5, max_iter 250)
    barrier()
    t1 = time.perf_counter()
    xs_m, _, reason_m = solve(martens=True)
    barrier()
    dt_m = time.perf_counter() - t1
    A_default = hf.DampedCurvature(op, 1.0)

    def solve_default():
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            return hf.cg(A_default, b, M=M, max_iter=250, martens_conv_crit=True, store_x_at_iters=None)

    default_leg = None
    try:  # (the headline must not depend on this leg; under data parallelism every rank takes the same branch: the
        # solver's kernels and the lockstep rule are deterministic, an exception here is a host-side one on all ranks)
        solve_default()  # (graph arguments of the new operator object: once, untimed)
        barrier()
        t1 = time.perf_counter()
        xs_d, _, reason_d = solve_default()
        barrier()
        dt_d = time.perf_counter() - t1
        default_leg = {"damping": 1.0, "iters": len(xs_d) - 1, "reason": reason_d, "iters_per_s": (len(xs_d) - 1) / dt_d,
                       "call": "cg(A, b, max_iter=250, martens_conv_crit=True, store_x_at_iters=None): the call "
                               "HessianFree.step makes (optimizer.py:265-274), wall time incl. A(x0), the backtracking "
                               "grid's snapshots and the final sync"}
    except Exception as exc:  # noqa: BLE001
        if group is not None:
            raise
        default_leg = {"error": repr(exc)}

    if rank == 0:
        k2_s = timing["k2_ms"] * 1e-3
        # K2 reads x,r,p,Bp,b(,minv), writes x,r; K1 8N, K3 12N (16N with minv)
        k2_bytes = (32.0 if M is not None else 28.0) * n
        it_bytes = (56.0 if M is not None else 48.0) * n
        achieved = k2_bytes / k2_s / 1e9 if k2_s > 0 else 0.0
        all3 = (timing["k1_ms"] + timing["k2_ms"] + timing["k3_ms"]) * 1e-3
        fused = bool(getattr(op, "_iteration_graphs", None))
        line = {
            "metric": "GGN-matvecs/sec (damped operator calls inside the PCG loop)",
            "value": world * matvecs / dt,
            "unit": "matvecs/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": f"{args.workload} {args.curvature.upper()} PCG solve: N={n} fp32 parameters, "
                            f"batch {args.batch}/GPU, {iters_done} PCG iterations in the last step "
                            f"({matvecs} operator calls in {args.steps} steps"
                            + (f" of {solves_per_step} Martens-terminated solves each" if timed_martens else "")
                            + f", max_iter {args.iters}), "
                            f"damping {args.damping}"
                            + (" (SURVEY.md 8(d) states 1.0: with 1.0 this random-init problem is at fp32 round-off "
                               "after ~15 iterations and 250 iterations cannot be sustained; `cg_to_martens` is the "
                               "Martens-terminated solve at THIS damping, `cg_default_damping` the solve at the "
                               "optimizer's default 1.0 -- which ends by tolerance, as the reference's --, and `step_ms` "
                               "runs complete default steps, damping 1.0 + LM)"
                               if args.damping != 1.0 and not hessian else "")
                            + f", {args.bn}-mode BN, CE-mean"
                            + (f" + L2 {l2:g}" if l2 > 0 else "") + ", x0=0, tol=0"
                            + (", diag empirical-Fisher preconditioner ^-0.75 (per-sample autograd)" if M is not None else "")
                            + (f", acc_step path: chunks {acc_sizes} accumulated" if acc_sizes else "")
                            + (f", FROZEN: {args.freeze} (requires_grad = False; N counts the trainable entries)"
                               if args.freeze else ""),
                "parallelism": f"dp{world} (batch sharded, {args.batch} samples per GPU, one all-reduce per matvec: the "
                               "4N-byte vector, or only its entries that can be non-zero with the fused engine -- "
                               "config.allreduce.bytes; `value` counts SHARD products: every rank's operator call "
                               f"counts once, so one product over the global batch of {world * args.batch} counts {world}x "
                               "-- weak scaling over the batch, SURVEY.md section 8e)"
                               + (f"; {world} ranks share {ndev} device(s) over gloo: functional run, not a "
                                  "scaling number" if oversubscribed else ""),
                "matvec": getattr(op, "mode", "eager autograd")
                          + ("; eval-BN fused (hf_chan_affine)" if args.fuse_bn else "")
                          + ("; conv tangent fused" if args.fuse_conv else "")
                          + "; convolutions: " + (os.environ.get("HF_CONV") or "auto") + "; " + layout
                          + ("; deterministic (two products bitwise equal)" if check.get("deterministic")
                             else "; NOT bitwise repeatable (library kernels with atomics)")
                          + ("; HessianFree.path_report(): "
                             + json.dumps({k: (v and {"path": v["path"], "declined": v["declined"]})
                                           for k, v in state["opt"].path_report().items()})
                             if state["opt"] is not None else ""),
                "iteration": ("one hipGraph launch per PCG iteration (product -> K1 -> K2 -> K3)" if fused and group is None
                              else "product graph A -> [all-reduce of the late layers' share on a second communicator] "
                                   "|| product graph B -> all-reduce of the rest -> scatter -> K1-K3 graph"
                              if fused and getattr(op, "split", None) is not None
                              else "product graph -> all-reduce -> K1-K3 graph" if fused
                              else "product, then K1, K2, K3 as separate launches"),
                "operator_source": ("HessianFree.acc_linearise(model, loss, chunks): what acc_step() itself hands to "
                                    "step() / cg() (accumulated engine session)" if state["opt"] is not None
                                    and getattr(state["opt"], "_acc_session", None) is op
                                    else "HessianFree.linearise(forward): the operator step() itself hands to cg() "
                                    "(persistent engine session)" if state["opt"] is not None
                                    and getattr(state["opt"], "_session", None) is op
                                    else "HessianFree.linearise(forward): the operator step() itself hands to cg() "
                                         "(generic path)" if state["opt"] is not None
                                    else "built by bench.py (--overlap 1, experimental)"),
                "path_report": (state["opt"].path_report() if state["opt"] is not None else None),
                "termination": reason,
                "allreduce": None if allreduce_ms is None else {
                    "path": comm_path, "bytes": int(getattr(getattr(op, "op", op), "reduce_bytes", 4 * n)),
                    "ms": allreduce_ms,
                    "backend": args.backend,
                    "world": world,
                    "ranks_seen": ranks_seen,
                    "ranks_seen_note": f"world {world}: ranks seen by the product's own collective path = {ranks_seen}",
                    "rung": int(os.environ.get("HF_BENCH_RUNG", "0")),
                    "rung_name": os.environ.get("HF_BENCH_RUNG_NAME", "(no ladder: single rank process)"),
                    "rungs_failed": json.loads(os.environ.get("HF_BENCH_RUNGS_FAILED", "[]")),
                    "validation": getattr(op, "mode_validation", None),
                    "side_stream_runs_beside_compute": getattr(op, "side_runs_beside", None),
                    "overlap_two_graphs": bool(args.overlap),
                    "product_mode": ("two-phase (chunked / overlapped)"
                                     if getattr(op, "split", None) is not None
                                     else "single graph + one compact all-reduce"),
                    "product_mode_timing_ms": getattr(op, "mode_timing", None),
                    "product_mode_policy": {-1: "auto (measured at session creation)", 0: "forced single",
                                            1: "forced two-phase"}.get(args.chunk)},
            },
            "cg_iters_per_s": world * args.steps * solves_per_step * iters_done / dt,
            "cg_to_martens": {"damping": args.damping, "iters": len(xs_m) - 1, "reason": reason_m,
                              "iters_per_s": (len(xs_m) - 1) / dt_m,
                              "call": "cg(A, b, max_iter=--iters, tol=0, martens_conv_crit=True, store_x_at_iters=[0])"},
            "cg_default_damping": default_leg,
            "roofline": {
                "bound": "hbm",
                "kernel": "k_update_xr (K2)",
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": None,
                "traffic_source": None,
                "alg_bytes_per_launch": k2_bytes,
                "avg_launch_ms": timing["k2_ms"],
                "launches_timed": timing["n"],
                "timing": "HIP events on the launch stream around K1/K2/K3"
                          + (f" (event-record nodes of every {_TIMING_SAMPLE}th iteration's graph)" if fused else
                             " of every iteration"),
                "all_pcg_kernels": {
                    "k1_ms": timing["k1_ms"], "k2_ms": timing["k2_ms"], "k3_ms": timing["k3_ms"],
                    "alg_bytes_per_iter": it_bytes,
                    "achieved_GBs": it_bytes / all3 / 1e9 if all3 > 0 else 0.0,
                },
            },
        }
        prof = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(prof):
            try:
                key = f"k_update_xr_{n}" + ("_precond" if M is not None else "")
                val = json.load(open(prof)).get(key)
                if val is not None:
                    line["roofline"]["traffic"] = val
                    line["roofline"]["traffic_source"] = (
                        "static: profiles/traffic.json (round 6 binary, scripts/collect_r06_profiles.sh) -- rocprofv3 "
                        "--pmc FETCH_SIZE / WRITE_SIZE passes taken on scripts/pcg_kernel_bench.py (the same kernel at "
                        "the same N; --pmc aborts this full command on this stack), not re-measured in this run")
            except Exception:
                pass
        if world == 1 and not dist_on and not args.no_beyond_l3:
            try:
                bc = beyond_cache_roofline(device)
                line["roofline"]["frac_beyond_l3"] = bc["k2_frac"]
                line["roofline"]["beyond_l3"] = bc
                line["roofline"]["note"] = (
                    f"frac: K2 at this workload's N = {n} (six {4 * n / 1e6:.1f} MB vectors = "
                    f"{6 * 4 * n / 2**20:.0f} MiB, at the edge of the 256 MiB Infinity Cache: partly served on-die); "
                    "frac_beyond_l3: the same kernel on 256 MiB vectors, measured in this run -- the plain HBM figure")
            except Exception as exc:  # noqa: BLE001
                line["roofline"]["beyond_l3"] = {"error": repr(exc)}
        if world == 1 and not dist_on and not args.no_step_timing:
            try:
                line["step_ms"] = full_step_timing(args, device)
            except Exception as exc:  # noqa: BLE001  (the headline number must not depend on this leg)
                line["step_ms"] = {"error": repr(exc)}
        if (world == 1 and not dist_on and not args.no_train_bn and args.bn == "eval" and args.workload == "resnet18"
                and args.curvature == "ggn" and not args.acc):
            line["train_bn"] = train_bn_leg(args)
        if world == 1 and not args.no_cpu_baseline:
            try:
                line["cpu_baseline"] = cpu_baseline(args)
            except Exception as exc:  # noqa: BLE001  (the headline must not depend on this leg)
                line["cpu_baseline"] = {"error": repr(exc)}
        print(json.dumps(line), flush=True)
    if group is not None:
        torch.distributed.barrier()
        marker = os.environ.get("HF_BENCH_DONE")
        if marker:  # (the run is complete on every rank: whatever the teardown does, this rung counts)
            with open(marker, "w") as fh:
                fh.write("done\n")
        torch.distributed.destroy_process_group()
        import faulthandler

        faulthandler.cancel_dump_traceback_later()


if __name__ == "__main__":
    main()

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
    ap.add_argument("--overlap", type=int, default=0,
                    help="data parallel: split the product into two hipGraphs and overlap the "
                         "all-reduce of the tail with the head's adjoint sweep (measured on one MI355X: "
                         "the split costs +0.19 ms and the two async collectives +0.4 ms per product, "
                         "so it only pays when the all-reduce itself takes > ~0.6 ms; off by default)")
    ap.add_argument("--force-dist", type=int, default=0,
                    help="create the process group even for WORLD_SIZE=1 (exercises the RCCL path on one GPU)")
    return ap.parse_args()


def build_problem(args, device, rank):
    from pytorchhessianfree_amd import testproblems as tp

    make = {"resnet18": tp.resnet18_mnist, "allcnnc": tp.allcnnc_cifar100,
            "resnet50": tp.resnet50_small_images}[args.workload]
    # same weights on every rank (seed 0), a different data shard per rank
    return make(batch_size=args.batch, seed=0, device=device, data_seed=1000 + rank)


def cpu_baseline(args):
    """The oracle (reference algorithm restated, torch CPU ops in the reference's
    order: oracle/pcg.py + oracle/backpack_restated.py) on the same workload,
    bounded to ``--cpu-iters`` PCG iterations."""
    from oracle import backpack_restated as bp
    from oracle import pcg as oracle
    from pytorchhessianfree_amd.utils import vector_to_parameter_list

    model, (x, t), lossf = build_problem(args, "cpu", 0)
    params = [p for p in model.parameters() if p.requires_grad]
    out = model(x)
    loss = lossf(out, t)
    grad = torch.cat([g.reshape(-1) for g in torch.autograd.grad(loss, params, retain_graph=True)])
    calls = [0]

    def mvp(v):  # optimizer.py:457-462 with BackPACK's published algorithm
        calls[0] += 1
        Gv = bp.ggn_vector_product_from_plist(loss, out, params, vector_to_parameter_list(v, params))
        return torch.cat([g.reshape(-1) for g in Gv]).detach()

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
        xs, _, _ = oracle.pcg(lambda v: mvp(v) + lam * v, -grad, max_iter=args.cpu_iters, tol=0.0,
                              martens_conv_crit=False, store_x_at_iters=[0])
    dt = time.perf_counter() - t0
    return {
        "value": calls[0] / dt,
        "unit": "GGN-matvecs/s",
        "cores": torch.get_num_threads(),
        "kind": "port",
        "sample": f"{len(xs)-1} PCG iterations ({calls[0]} matvecs) of the same {args.workload} "
                  f"batch-{args.batch} problem, {dt:.1f} s",
        "cg_iters_per_s": (len(xs) - 1) / dt,
        "host_logical_cores": os.cpu_count(),
    }


def main():
    args = parse()
    if args.channels_last < 0:
        args.channels_last = int(args.workload in ("resnet18", "allcnnc"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an AMD GPU: the hot path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    import pytorchhessianfree_amd as hf
    from pytorchhessianfree_amd import curvature, modelprep
    from pytorchhessianfree_amd.cg import enable_kernel_timing, read_kernel_timing

    weight = 1.0 / world

    def build_operator(channels_last):
        model, (x, t), lossf = build_problem(args, device, rank)
        if args.fuse_bn:
            modelprep.fuse_eval_batchnorm(model)
        if args.fuse_conv:
            modelprep.fuse_conv_tangent(model, channels_last=channels_last)
        if args.fuse_bn and args.fuse_conv:
            modelprep.fuse_residual_blocks(model)  # relu(bn(.)) / relu(bn(.) + identity) as one layer
            modelprep.fuse_bn_relu(model)          # relu(bn(.)) outside residual blocks (the stem)
            modelprep.skip_identity_pools(model)   # AdaptiveAvgPool2d(1) of a 1x1 map
        params = [p for p in model.parameters() if p.requires_grad]
        # local gradient first (its graph is freed again: nothing may tie the parameters
        # to the default stream while the product is captured, see GraphedOperator)
        grad = curvature.flatten_into(torch.autograd.grad(lossf(model(x), t), params), params,
                                      scale=weight)

        def builder():  # forward graph + recorded J^T / H_L maps (once per Newton step)
            out = model(x)
            return curvature.GGNOperator(lossf(out, t), out, params, weight=weight, group=None)

        # The local product is captured BEFORE the process group exists: RCCL's
        # watchdog thread must not touch the runtime while a capture is open.
        if args.graph and args.overlap and (world > 1 or args.force_dist):
            op = curvature.OverlappedGraphedOperator(builder, params=params)
        else:
            op = curvature.maybe_graphed(builder, enable=bool(args.graph), params=params)
        return op, grad, sum(p.numel() for p in params)

    def stock_product(v, dtype):
        """The same product by stock PyTorch-ROCm autograd on an unpatched NCHW model
        (float64: the reference; float32: what the stock fp32 path itself achieves)."""
        model, (x, t), lossf = build_problem(args, device, rank)
        model, x = model.to(dtype), x.to(dtype)
        params = [p for p in model.parameters() if p.requires_grad]
        find = torch.backends.cudnn.benchmark
        # one product only: MIOpen's immediate mode, no find step (and no find-db records)
        # for the shapes of PyTorch's generic double-backward
        torch.backends.cudnn.benchmark = False
        try:
            out = model(x)
            op = curvature.GGNOperator(lossf(out, t), out, params, weight=weight, group=None)
            return op(v.to(dtype)).double().clone()
        finally:
            torch.backends.cudnn.benchmark = find

    # guard: an operator is only timed after it reproduced a float64 stock-autograd product
    # as well as stock fp32 autograd does (x5; deep random-init nets such as the ResNet-50
    # workload are only good to ~1e-4 in fp32, ResNet-18 to 2e-7).  A wrong operator must
    # never be what gets measured; NHWC falls back to NCHW, NCHW aborts.
    check = {}

    def checked(channels_last):
        op, grad, n = build_operator(channels_last)
        if "want" not in check:
            v = torch.randn(n, device=device, generator=torch.Generator(device=device).manual_seed(7))
            want = stock_product(v, torch.float64)
            scale = float(want.abs().max())
            stock_err = float((stock_product(v, torch.float32) - want).abs().max()) / scale
            check.update(v=v, want=want, scale=scale, stock_err=stock_err, tol=max(1e-5, 5.0 * stock_err))
        err = float((op(check["v"]).double() - check["want"]).abs().max()) / check["scale"]
        return op, grad, n, err

    op, grad, n, err = checked(bool(args.channels_last))
    note = "max-norm error against float64 stock autograd {:.1e}, stock fp32 autograd {:.1e}"
    layout = ("NHWC" if args.channels_last else "NCHW") + " (" + note.format(err, check["stock_err"]) + ")"
    if not err < check["tol"] and args.channels_last:
        print(f"[bench] NHWC product off by {err:.2e} (float64 reference); using NCHW",
              file=sys.stderr, flush=True)
        del op
        op, grad, n, err = checked(False)
        layout = "NCHW (NHWC failed its check; " + note.format(err, check["stock_err"]) + ")"
    if not err < check["tol"]:
        raise SystemExit(f"bench: the curvature product is off by {err:.2e} against float64 stock autograd")
    del check["want"], check["v"]

    group = None
    if world > 1 or args.force_dist:
        import torch.distributed as dist

        if world == 1:  # --force-dist on a single GPU, started without a launcher
            for key, val in (("RANK", "0"), ("WORLD_SIZE", "1"), ("MASTER_ADDR", "127.0.0.1"),
                             ("MASTER_PORT", "29517")):
                os.environ.setdefault(key, val)
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(args.backend)
        group = dist.group.WORLD
    op.group = group  # one all-reduce (sum) of the 4N-byte partial product per matvec

    if group is not None:
        torch.distributed.all_reduce(grad, group=group)
    b = -grad
    A = hf.DampedCurvature(op, args.damping)

    def solve(martens=False, max_iter=args.iters):
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            return hf.cg(A, b, max_iter=max_iter, tol=0.0, martens_conv_crit=martens,
                         store_x_at_iters=[0])

    def barrier():
        if group is not None:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        solve()
    ws = enable_kernel_timing(device, n, torch.float32, not os.environ.get("HF_BENCH_NO_TIMING"))
    calls0 = op.calls
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        xs, _, reason = solve()
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

    # CG-iters/sec to Martens' criterion (second half of BASELINE.json's metric)
    barrier()
    t1 = time.perf_counter()
    xs_m, _, reason_m = solve(martens=True)
    barrier()
    dt_m = time.perf_counter() - t1

    if rank == 0:
        k2_s = timing["k2_ms"] * 1e-3
        alg_bytes = 28.0 * n  # K2: reads x,r,p,Bp,b, writes x,r (fp32)  SURVEY.md 8(d)
        achieved = alg_bytes / k2_s / 1e9 if k2_s > 0 else 0.0
        all3 = (timing["k1_ms"] + timing["k2_ms"] + timing["k3_ms"]) * 1e-3
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
                "workload": f"{args.workload} GGN PCG solve: N={n} fp32 parameters, "
                            f"batch {args.batch}/GPU, {iters_done} PCG iterations/step, "
                            f"damping {args.damping}, eval-mode BN, CE-mean, x0=0, tol=0",
                "parallelism": f"dp{world} (batch sharded, one all-reduce of 4N bytes per matvec)",
                "matvec": getattr(op, "mode", "eager autograd")
                          + ("; eval-BN fused (hf_chan_affine)" if args.fuse_bn else "")
                          + ("; conv tangent fused" if args.fuse_conv else "") + "; " + layout,
                "termination": reason,
            },
            "cg_iters_per_s": world * args.steps * iters_done / dt,
            "cg_to_martens": {"iters": len(xs_m) - 1, "reason": reason_m,
                              "iters_per_s": (len(xs_m) - 1) / dt_m},
            "roofline": {
                "bound": "hbm",
                "kernel": "k_update_xr (K2)",
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": None,
                "alg_bytes_per_launch": alg_bytes,
                "avg_launch_ms": timing["k2_ms"],
                "launches_timed": timing["n"],
                "all_pcg_kernels": {
                    "k1_ms": timing["k1_ms"], "k2_ms": timing["k2_ms"], "k3_ms": timing["k3_ms"],
                    "alg_bytes_per_iter": 48.0 * n,
                    "achieved_GBs": 48.0 * n / all3 / 1e9 if all3 > 0 else 0.0,
                },
            },
        }
        prof = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(prof):
            try:
                line["roofline"]["traffic"] = json.load(open(prof)).get(f"k_update_xr_{n}")
            except Exception:
                pass
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(args)
        print(json.dumps(line), flush=True)
    if group is not None:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()

"""Micro-benchmark of the three PCG kernels (K1 curvature, K2 update_xr, K3 update_p)
on a synthetic diagonal operator at BASELINE.json's vector lengths.

    python scripts/pcg_kernel_bench.py [--sizes 1387108,11175370,25557032] [--precond 0|1]

Durations are HIP-event timed inside libhfpcg (hf_pcg_timing_*), achieved GB/s is
algorithmic bytes / duration.  Prints one JSON line per size.
"""

import argparse
import json
import os
import sys
import warnings

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pytorchhessianfree_amd as hf  # noqa: E402
from pytorchhessianfree_amd.cg import enable_kernel_timing, read_kernel_timing  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sizes", default="1387108,11175370,25557032,100000000")
    ap.add_argument("--precond", type=int, default=0)
    ap.add_argument("--iters", type=int, default=200)
    args = ap.parse_args()
    dev = torch.device("cuda")
    for n in [int(s) for s in args.sizes.split(",")]:
        gen = torch.Generator(device=dev).manual_seed(0)
        d = torch.rand(n, device=dev, generator=gen) * 100.0 + 1e-3
        b = torch.randn(n, device=dev, generator=gen)
        out = torch.empty(n, device=dev)

        def B(v):
            return torch.mul(d, v, out=out)

        A = hf.DampedCurvature(B, 1e-3)
        M = hf.DiagonalPreconditioner(torch.rand(n, device=dev, generator=gen), 1e-3) if args.precond else None
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            hf.cg(A, b, M=M, max_iter=20, tol=0.0, store_x_at_iters=[0])  # warm-up
            ws = enable_kernel_timing(dev, n, torch.float32, True)
            xs, _, reason = hf.cg(A, b, M=M, max_iter=args.iters, tol=0.0, store_x_at_iters=[0])
        t = read_kernel_timing(ws)
        enable_kernel_timing(dev, n, torch.float32, False)
        by = {"k1": 8, "k2": 32 if args.precond else 28, "k3": 16 if args.precond else 12}
        line = {"n": n, "precond": args.precond, "iters": len(xs) - 1, "reason": reason}
        tot_b = tot_t = 0.0
        for k in ("k1", "k2", "k3"):
            ms = t[k + "_ms"]
            line[k + "_us"] = round(ms * 1e3, 2)
            line[k + "_GBs"] = round(by[k] * n / (ms * 1e-3) / 1e9, 1)
            tot_b += by[k] * n
            tot_t += ms * 1e-3
        line["all_GBs"] = round(tot_b / tot_t / 1e9, 1)
        line["frac_of_8TBs"] = round(tot_b / tot_t / 8e12, 3)
        print(json.dumps(line), flush=True)
        del d, b, out, xs
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()

"""Worst observed value / stated bound per ``tol.within`` site over several runs of the GPU suite
(``HF_TOL_LOG=gpurun_out/<lease>/tol.jsonl python -m pytest tests -m gpu``): lists the sites whose bound is less than
``--margin`` (default 3) times the worst value seen on any lease.  Integer bounds (iteration counts) are skipped.

    python scripts/tolerance_report.py gpurun_out/r5*/tol*.jsonl
"""
import argparse
import json

ap = argparse.ArgumentParser()
ap.add_argument("logs", nargs="+")
ap.add_argument("--margin", type=float, default=3.0)
args = ap.parse_args()
worst = {}
for path in args.logs:
    for line in open(path):
        r = json.loads(line)
        if float(r["bound"]).is_integer() and r["bound"] >= 1:
            continue
        w = worst.get(r["site"])
        if w is None or r["worst_ratio"] > w["worst_ratio"]:
            worst[r["site"]] = dict(r, log=path)
tight = sorted((w for w in worst.values() if w["worst_ratio"] * args.margin > 1.0), key=lambda w: -w["worst_ratio"])
print(f"{len(worst)} sites, {len(tight)} with bound < {args.margin} x worst observed")
for w in tight:
    print(f"{w['site']:32s} worst {w['worst_value']:.3e}  bound {w['bound']:.3e}  ratio {w['worst_ratio']:.2f}  -> bound >= "
          f"{args.margin * w['worst_value']:.1e}   {w['test']}")

#!/bin/bash
# (experiment) where the 0.665 ms of a PCG iteration are: kernel trace of a short bench run; per iteration (K3 end -> next
# K3 end) the sum of kernel durations and the idle time between kernels, split into "inside the graph" and "between two
# graph launches" (K3 of iteration i -> first kernel of iteration i+1).
OUT=${1:-gpurun_out/gap}; mkdir -p $OUT; export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d /tmp/gap_trace -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-step-timing --no-beyond-l3 --no-train-bn > $OUT/bench.json 2> $OUT/err.log
python3 - /tmp/gap_trace $OUT <<'PY'
import csv, glob, json, os, sys
rows = []
for p in glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True):
    rows += list(csv.DictReader(open(p)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
k3 = [i for i, r in enumerate(rows) if "k_update_p" in r["Kernel_Name"]]
its = []
for a, b in zip(k3[200:700], k3[201:701]):  # iterations well inside a 250-iteration solve
    seg = rows[a + 1: b + 1]
    if len(seg) < 60 or len(seg) > 90:
        continue
    t_prev_end = int(rows[a]["End_Timestamp"])
    busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in seg)
    first_gap = int(seg[0]["Start_Timestamp"]) - t_prev_end
    inner_gap = sum(int(y["Start_Timestamp"]) - int(x["End_Timestamp"]) for x, y in zip(seg, seg[1:]))
    total = int(seg[-1]["End_Timestamp"]) - t_prev_end
    its.append((total, busy, first_gap, inner_gap, len(seg)))
n = len(its)
avg = [sum(v[i] for v in its) / n / 1e3 for i in range(4)]
out = {"iterations": n, "launches_per_iteration": its[0][4], "us_per_iteration": avg[0], "kernel_busy_us": avg[1],
       "gap_between_graph_launches_us": avg[2], "gaps_inside_the_graph_us": avg[3]}
print(json.dumps(out))
json.dump(out, open(os.path.join(sys.argv[2], "iteration_gap.json"), "w"), indent=1)
PY
rm -rf /tmp/gap_trace

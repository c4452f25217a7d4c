#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf gpurun_out/pmc_$C
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d gpurun_out/pmc_$C -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --graph 0 --iters 12 > gpurun_out/pmc_$C.log 2>&1
  echo "== $C rc=$?"; tail -2 gpurun_out/pmc_$C.log | cut -c1-300
  ls gpurun_out/pmc_$C/*/ | head
done
python3 - <<'PY'
import csv, glob, collections
out = {}
for C in ("FETCH_SIZE", "WRITE_SIZE"):
    fs = glob.glob(f"gpurun_out/pmc_{C}/*/*counter_collection.csv")
    print(C, fs)
    if not fs: continue
    acc = collections.defaultdict(list)
    with open(fs[0]) as f:
        rd = csv.DictReader(f)
        for row in rd:
            name = row.get("Kernel_Name", "")
            if "k_update" in name or "k_curv" in name or "k_pack" in name or "k_unpack" in name:
                key = name.split("(")[0].split("::")[-1][:40]
                if row.get("Counter_Name") == C:
                    acc[key].append(float(row["Counter_Value"]))
    for k, v in acc.items():
        print(C, k, "n", len(v), "mean", sum(v) / len(v), "min", min(v), "max", max(v))
        out[(C, k)] = sum(v) / len(v)
import json
json.dump({f"{c}|{k}": v for (c, k), v in out.items()}, open("gpurun_out/pmc_summary.json", "w"), indent=1)
PY
find gpurun_out/pmc_* -name "*kernel_trace.csv" -size +5M -delete; find gpurun_out/pmc_* -name "*counter_collection.csv" -size +20M -delete

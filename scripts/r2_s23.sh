cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/s23; mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_k -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $R/$O/prof_bench.json 2> $R/$O/prof_bench.err)
f=$(find /tmp/prof_k -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/kernel_stats.csv
python - <<PY
import csv
rows=list(csv.DictReader(open("$O/kernel_stats.csv")))
tot=sum(float(r['TotalDurationNs']) for r in rows)
for r in rows[:24]:
    print(f"{r['Name'][:70]:70s} {int(r['Calls']):7d} {float(r['AverageNs'])/1e3:8.2f} {100*float(r['TotalDurationNs'])/tot:6.2f}")
PY
for v in "HF_BN_ROW_BLOCKS=64" "HF_BN_ROW_PASSES=1" "HF_BN_ROW_BLOCKS=64 HF_BN_ROW_PASSES=1" "HF_BN_ROW_BLOCKS=16"; do
  echo "== $v"; env $v timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
r = json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(r['value'],1), r['ms_per_step'])"
done
timeout 900 python bench.py --workload resnet50 --channels-last 1 --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_resnet50_cl.json 2>$O/bench_resnet50_cl.err; echo "resnet50 cl rc=$?"; tail -c 400 $O/bench_resnet50_cl.err; python -c "
import json; r=json.loads(open('$O/bench_resnet50_cl.json').read().strip().splitlines()[-1]); print(round(r['value'],1), r['config']['matvec'])"

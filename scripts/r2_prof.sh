# rocprofv3 per-kernel summary of the bench command (fresh output directory: boxes may be reused)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r2final4; mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
D=/tmp/prof_k_$$; rm -rf $D
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $D -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $R/$O/prof_bench.json 2> $R/$O/prof_bench.err)
f=$(find $D -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/kernel_stats_full.csv
python - <<PY
import csv
rows=list(csv.DictReader(open("$O/kernel_stats_full.csv")))
for r in rows[:26]:
    print(f"{r['Name'][:70]:70s} {int(r['Calls']):7d} {float(r['AverageNs'])/1e3:8.2f}")
PY

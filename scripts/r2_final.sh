# round 2 measurement session (PMC traffic is taken separately: scripts/r2_pmc.sh)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r2final4; mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | grep -v amdgpu.ids | tail -3
timeout 900 python bench.py > $O/bench_n1.json 2> $O/bench_n1.err; echo "bench rc=$?"
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_k_$$ -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $R/$O/prof_bench.json 2> $R/$O/prof_bench.err)
f=$(find /tmp/prof_k_$$ -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/kernel_stats_full.csv
timeout 300 python scripts/conv_kernel_bench.py --no-reduce 1 > $O/conv_kernel_bench.jsonl 2>/dev/null
rm -f $O/conv_modes.jsonl
timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline >> $O/conv_modes.jsonl 2>/dev/null
for m in own miopen; do timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --engine 0 --conv $m >> $O/conv_modes.jsonl 2>/dev/null; done
timeout 900 python bench.py --gpus 2 --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_2ranks_one_gpu.json 2> $O/bench_2ranks.err; echo "2rank rc=$?"
rm -f $O/other_workloads.jsonl
timeout 900 python bench.py --workload resnet50 --steps 2 --warmup 1 --no-cpu-baseline >> $O/other_workloads.jsonl 2>$O/bench_resnet50.err; echo "resnet50 rc=$?"; tail -c 300 $O/bench_resnet50.err
timeout 900 python bench.py --workload allcnnc --steps 2 --warmup 1 --no-cpu-baseline >> $O/other_workloads.jsonl 2>/dev/null; echo "allcnnc rc=$?"
timeout 900 python bench.py --workload allcnnc --curvature hessian --precond 1 --damping 1.0 --steps 2 --warmup 1 >> $O/other_workloads.jsonl 2>/dev/null; echo "config4 rc=$?"
timeout 900 python bench.py --force-dist 1 --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_dist1.json 2>/dev/null; echo "dist1 rc=$?"
timeout 300 python scripts/pcg_kernel_bench.py > $O/pcg_kernel_bench.jsonl 2>/dev/null; timeout 300 python scripts/pcg_kernel_bench.py --precond 1 >> $O/pcg_kernel_bench.jsonl 2>/dev/null
python - <<PY
import json,glob
for f in sorted(glob.glob("$O/bench*.json")) + ["$O/other_workloads.jsonl", "$O/conv_modes.jsonl"]:
    try:
        for l in open(f).read().strip().splitlines():
            if not l.startswith("{"): continue
            r=json.loads(l); print(f.split("/")[-1], round(r["value"],1), r["n_gpus"], round(r["roofline"]["frac"],3), r["config"]["termination"], r["config"]["matvec"][:60])
    except Exception as e: print(f, "ERR", e)
PY

# round 2 measurement session (PMC traffic is taken separately: scripts/r2_pmc.sh)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r2final; mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 300 python scripts/pcg_kernel_bench.py > $O/pcg_kernel_bench.jsonl 2>/dev/null
timeout 300 python scripts/pcg_kernel_bench.py --precond 1 >> $O/pcg_kernel_bench.jsonl 2>/dev/null
timeout 300 python scripts/conv_kernel_bench.py > $O/conv_kernel_bench.jsonl 2>/dev/null
for m in own miopen; do timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --conv $m >> $O/conv_modes.jsonl 2>/dev/null; done
timeout 900 python bench.py --gpus 2 --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_2ranks_one_gpu.json 2> $O/bench_2ranks.err; echo "2rank rc=$?"
timeout 900 python bench.py --workload allcnnc --curvature hessian --precond 1 --damping 1.0 --steps 3 --warmup 1 > $O/bench_config4.json 2>/dev/null
timeout 900 python bench.py --workload allcnnc --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_allcnnc_ggn.json 2>/dev/null
timeout 900 python bench.py --workload resnet50 --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_resnet50.json 2>/dev/null
(cd /tmp && timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pmcx -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --graph 0 --iters 12 --fuse-bn 0 --fuse-conv 0 > /dev/null 2> $R/$O/pmc_stock.err; echo "pmc stock-layers rc=$?")
timeout 2400 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest.log
python - <<PY
import json,glob
for f in sorted(glob.glob("$O/bench*.json")):
    try:
        r=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split("/")[-1], round(r["value"],1), r["n_gpus"], round(r["roofline"]["frac"],3), r["config"]["termination"])
    except Exception as e: print(f, "ERR", e)
PY
cat $O/pcg_kernel_bench.jsonl

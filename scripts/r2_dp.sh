# data-parallel checks on one GPU: kernels, 2 ranks over gloo, 1-rank RCCL group
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r2final3; mkdir -p $O
timeout 1500 python -m pytest tests/test_engine_gpu.py tests/test_distributed_gpu.py -q -x -p no:cacheprovider -k "live_copy or two_ranks or rccl or engine_product" 2>&1 | grep -v amdgpu.ids | tail -4
timeout 900 python bench.py --force-dist 1 --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_dist1.json 2>/dev/null; echo "dist1 rc=$?"
timeout 900 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_n1_short.json 2>/dev/null
python - <<PY
import json
for f in ("bench_dist1.json", "bench_n1_short.json"):
    r=json.loads(open("$O/"+f).read().strip().splitlines()[-1]); print(f, round(r["value"],1), r["config"]["allreduce"])
PY

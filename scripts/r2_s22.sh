cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/s22; mkdir -p $O
timeout 900 python -m pytest tests/test_engine_gpu.py tests/test_boundary.py -q -x -p no:cacheprovider 2>&1 | grep -v amdgpu.ids | tail -15
for v in "" "HF_ENGINE_HEAD=0" "HF_ENGINE_POOL=0" "HF_ENGINE_LIVE=0"; do
  echo "== $v"; env $v timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
r = json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(r['value'],1), r['ms_per_step'], r['config']['matvec'][-60:])"
done
timeout 600 python bench.py --workload resnet50 --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_resnet50.json 2>$O/bench_resnet50.err; echo "resnet50 rc=$?"; tail -c 300 $O/bench_resnet50.err

cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/s25; mkdir -p $O
timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
r = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench', round(r['value'],1), r['ms_per_step'])"
timeout 2400 python -m pytest tests -m gpu -q -p no:cacheprovider > $O/pytest.log 2>&1; echo "pytest rc=$?"; grep -v "^  File\|amdgpu.ids" $O/pytest.log | tail -15
timeout 600 python bench.py --workload resnet50 --steps 2 --warmup 1 --no-cpu-baseline 2>$O/r50.err | python -c "
import sys, json
r = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('resnet50', round(r['value'],1), r['config']['matvec'][-160:])"; tail -c 300 $O/r50.err

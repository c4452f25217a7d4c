cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r2s17; mkdir -p $O
timeout 600 python -m pytest tests/test_engine_gpu.py -m gpu -q -k bottleneck 2>&1 | grep -v "^  File\|amdgpu.ids" | grep "passed\|failed\|AssertionError" | head
timeout 1500 python -m pytest tests/test_engine_gpu.py tests/test_optimizer_gpu.py -m gpu -q -x 2>&1 | grep -v "^  File\|amdgpu.ids" | tail -8
timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/b.json 2> $O/b.err || tail -c 400 $O/b.err
python -c "
import json; r=json.load(open('$O/b.json')); print('bench', round(r['value'],1), r['cg_to_martens'], r['config']['matvec'][-150:])"

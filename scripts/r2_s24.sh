cd "$GRAFT_REPO_ROOT" || exit 1
timeout 900 python -m pytest tests/test_engine_gpu.py -q -x -p no:cacheprovider -k "head or maxpool or resnet18_engine" 2>&1 | grep -v amdgpu.ids | tail -8
for v in "HF_BN_ROW_BLOCKS=64 HF_BN_ROW_PASSES=1" "HF_BN_ROW_BLOCKS=128 HF_BN_ROW_PASSES=1" "HF_BN_ROW_BLOCKS=256 HF_BN_ROW_PASSES=1" "HF_BN_ROW_BLOCKS=96 HF_BN_ROW_PASSES=1"; do
  echo "== $v"; env $v timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
r = json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(r['value'],1), r['ms_per_step'])"
done

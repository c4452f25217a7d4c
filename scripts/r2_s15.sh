cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r2s15; mkdir -p $O
timeout 900 python -m pytest tests/test_engine_gpu.py tests/test_conv_gpu.py -m gpu -q -x 2>&1 | grep -v "^  File\|amdgpu.ids" | tail -25
for b in 0 256 384 768; do
  HF_CONV_BLOCKS=$b timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/b_$b.json 2> $O/b_$b.err || tail -c 400 $O/b_$b.err
  python - <<PY
import json
r=json.load(open("$O/b_$b.json")); print("blocks=$b", round(r["value"],1), r["config"]["matvec"][:60], "|", r["config"]["matvec"][-60:])
PY
done
timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --engine 0 > $O/b_noengine.json 2>/dev/null; python -c "
import json; r=json.load(open('$O/b_noengine.json')); print('engine=0', round(r['value'],1))"

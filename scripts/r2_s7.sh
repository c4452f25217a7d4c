cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r2s7; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_conv_gpu.py -m gpu -q -x 2>&1 | tail -5
for m in auto own miopen; do
  timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --conv $m > $O/bench_$m.json 2> $O/bench_$m.err; echo "bench $m rc=$?"
  tail -c 400 $O/bench_$m.err
  python - <<PY
import json
r=json.load(open("$O/bench_$m.json"))
print("$m", round(r["value"],1), r["config"]["matvec"][-140:])
PY
done
python scripts/experiments/determinism_probe.py 2>&1 | grep -v amdgpu.ids | head -8
HF_CONV=own python scripts/experiments/determinism_probe.py 2>&1 | grep -v amdgpu.ids | head -8

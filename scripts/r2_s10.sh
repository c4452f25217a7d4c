cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r2s10; mkdir -p $O
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc=$?"
grep -v "^  File\|amdgpu.ids" $O/pytest.log | tail -40
timeout 600 python bench.py --force-dist 1 --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_dist1.json 2> $O/bench_dist1.err; echo "rc=$?"; tail -c 500 $O/bench_dist1.err
python - <<PY
import json
r=json.load(open("$O/bench_dist1.json")); print(round(r["value"],1), r["config"]["iteration"], r["config"]["allreduce"])
PY

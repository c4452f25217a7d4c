cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r2s8; mkdir -p $O
export TMPDIR=/tmp
for spec in "DW:32" "DW:32,W:128" "DW:32,T:32" "DW:32,stem:1" "DW:128" "W:128" "DW:0"; do
  HF_CONV_AUTO="$spec" timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/b.json 2> $O/b.err || tail -c 300 $O/b.err
  python - <<PY
import json
r=json.load(open("$O/b.json"))
print("$spec", round(r["value"],1))
PY
done

cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r2s16; mkdir -p $O
timeout 900 python -m pytest tests/test_engine_gpu.py -m gpu -q 2>&1 | grep -v "^  File\|amdgpu.ids" | grep -B2 -A12 "Error\|assert" | head -60
for i in 1 2; do
timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/b.json 2> $O/b.err || tail -c 400 $O/b.err
python -c "
import json; r=json.load(open('$O/b.json')); print('bench', round(r['value'],1), r['cg_to_martens'])"
done
bash scripts/r2_s13.sh 2>&1 | tail -22

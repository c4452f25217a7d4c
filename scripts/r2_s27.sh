cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r2final3; mkdir -p $O
timeout 900 python -m pytest tests/test_engine_gpu.py -q -p no:cacheprovider 2>&1 | grep -v amdgpu.ids | tail -3
timeout 900 python bench.py > $O/bench_n1.json 2> $O/bench_n1.err; echo "bench rc=$?"
python -c "
import json; r=json.loads(open('$O/bench_n1.json').read().strip().splitlines()[-1]); print(round(r['value'],1), r['ms_per_step'], r['roofline']['frac'], r['roofline']['avg_launch_ms'], r['roofline']['all_pcg_kernels'])"
bash scripts/r2_prof.sh | grep "head\|maxpool\|k_pack\|update_xr"
timeout 900 python bench.py --gpus 2 --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_2ranks_one_gpu.json 2> $O/bench_2ranks.err; echo "2rank rc=$?"; head -c 300 $O/bench_2ranks_one_gpu.json

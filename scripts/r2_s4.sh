set -x
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r2s4; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_conv_gpu.py -m gpu -q -x > $O/pytest_conv.log 2>&1; echo "pytest conv rc=$?"
tail -5 $O/pytest_conv.log
timeout 300 python scripts/conv_kernel_bench.py > $O/convbench.jsonl 2>&1
timeout 300 python scripts/conv_kernel_bench.py --blocks 128 > $O/convbench_128.jsonl 2>&1
timeout 300 python scripts/conv_kernel_bench.py --blocks 256 > $O/convbench_256.jsonl 2>&1
cat $O/convbench.jsonl; echo; cat $O/convbench_128.jsonl; echo; cat $O/convbench_256.jsonl
timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_own.json 2> $O/bench_own.err; echo "bench rc=$?"
tail -c 800 $O/bench_own.err
cut -c1-1300 $O/bench_own.json

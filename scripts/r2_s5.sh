set -x
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r2s5; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_conv_gpu.py -m gpu -q -x > $O/pytest_conv.log 2>&1; echo "pytest conv rc=$?"
tail -3 $O/pytest_conv.log
timeout 300 python scripts/conv_kernel_bench.py > $O/convbench.jsonl 2>&1
timeout 300 python scripts/conv_kernel_bench.py --blocks 256 > $O/convbench_256.jsonl 2>&1
cat $O/convbench.jsonl; echo; cat $O/convbench_256.jsonl

set -x
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r2s2; mkdir -p $O
export TMPDIR=/tmp
python scripts/experiments/gemm_shapes.py > $O/gemm_default.jsonl 2>&1
python scripts/experiments/gemm_shapes.py hipblaslt > $O/gemm_lt.jsonl 2>&1
python scripts/experiments/gemm_shapes.py rocblas > $O/gemm_rocblas.jsonl 2>&1
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_gemm -- python3 $GRAFT_REPO_ROOT/scripts/experiments/gemm_shapes.py > /dev/null 2>&1)
f=$(find /tmp/prof_gemm -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cut -c1-260 "$f" | head -60 > $O/gemm_kernel_stats.csv
cat $O/gemm_default.jsonl $O/gemm_lt.jsonl $O/gemm_rocblas.jsonl

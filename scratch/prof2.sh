#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/ -m gpu -q 2>&1 | tail -5
echo "=== bench eager"; timeout 600 python bench.py --graph 0 --steps 2 --warmup 1 --no-cpu-baseline
echo "=== bench graph"; timeout 900 python bench.py --steps 3 --warmup 1 | tee gpurun_out/bench_r1_n1.json
rm -rf gpurun_out/prof_r1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r1 -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/prof_r1.log 2>&1
f=$(find gpurun_out/prof_r1 -name "*kernel_stats.csv" | head -1); echo $f; grep -E "k_update|k_curv|k_pack|k_init" "$f" | cut -c1-200
find gpurun_out/prof_r1 -name "*kernel_trace.csv" -size +30M -delete

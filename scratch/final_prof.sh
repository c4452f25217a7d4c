#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/ -m gpu -q 2>&1 | tail -3
rm -rf gpurun_out/prof_final
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_final -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/prof_final.log 2>&1
tail -1 gpurun_out/prof_final.log | cut -c1-250
find gpurun_out/prof_final -name "*kernel_trace.csv" -delete

#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof5
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof5 -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/prof5.log 2>&1
tail -1 gpurun_out/prof5.log | cut -c1-200
find gpurun_out/prof5 -name "*kernel_trace.csv" -delete

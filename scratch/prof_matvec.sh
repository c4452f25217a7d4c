#!/bin/bash
# kernel trace of the bench (eager matvec) -> per-kernel stats
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof1 -- python3 bench.py --graph 0 --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/prof1.log 2>&1
find gpurun_out/prof1 -name "*kernel_stats*" | head
f=$(find gpurun_out/prof1 -name "*kernel_stats.csv" | head -1)
head -40 "$f" | cut -c1-220
# keep only the stats (trace csv is big)
find gpurun_out/prof1 -name "*kernel_trace.csv" -size +20M -delete

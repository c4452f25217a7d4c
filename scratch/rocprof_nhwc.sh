#!/bin/bash
# does the NHWC check fail under rocprofv3 when no MIOpen find step runs in the process?
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for i in 1 2 3 4 5; do
rm -rf gpurun_out/prof_t$i
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_t$i -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/prof_t$i.log 2>&1
echo "immediate $i: $(grep -a '\[bench\]' gpurun_out/prof_t$i.log) $(grep -a -o '"value": [0-9.]*' gpurun_out/prof_t$i.log) $(grep -a -c naive gpurun_out/prof_t$i/*/*kernel_stats.csv)"
find gpurun_out/prof_t$i -name "*kernel_trace.csv" -delete
done
export HF_BENCH_STOCK_FIND=1
for i in 6 7 8; do
rm -rf gpurun_out/prof_t$i
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_t$i -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/prof_t$i.log 2>&1
echo "find $i: $(grep -a '\[bench\]' gpurun_out/prof_t$i.log) $(grep -a -o '"value": [0-9.]*' gpurun_out/prof_t$i.log) $(grep -a -c naive gpurun_out/prof_t$i/*/*kernel_stats.csv)"
find gpurun_out/prof_t$i -name "*kernel_trace.csv" -delete
done

#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
show() { grep -a "\[bench\]" $1; grep -a '^{"metric"' $1 | tail -1 | python -c "
import sys,json
try:
    d=json.loads(sys.stdin.read()); print(round(d['value'],1), d['config']['workload'][:30], '|', d['config']['matvec'][-110:], round(d['roofline']['frac'],3))
except Exception as e: print('no json', e)"; }
for i in 1 2; do
timeout 900 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r14_$i.log 2>&1; show gpurun_out/r14_$i.log
done
rm -rf gpurun_out/prof_r14
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r14 -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/prof_r14.log 2>&1
grep -a -E "k_chan_affine" gpurun_out/prof_r14/*/*kernel_stats.csv | cut -d, -f1-4 | cut -c1-150
find gpurun_out/prof_r14 -name "*kernel_trace.csv" -delete

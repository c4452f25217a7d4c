#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
show() { grep -a "\[bench\]" $1; grep -a '^{"metric"' $1 | tail -1 | python -c "
import sys,json
try:
    d=json.loads(sys.stdin.read()); print(round(d['value'],1), d['config']['workload'][:30], '|', d['config']['matvec'][-110:], round(d['roofline']['frac'],3))
except Exception as e: print('no json', e)"; }
timeout 1800 python -m pytest tests/ -m gpu -q 2>&1 | tail -3
timeout 600 python __graft_entry__.py smoke 2>&1 | tail -1
timeout 1200 python bench.py > gpurun_out/bench_r1_final8.json 2> gpurun_out/bench_r1_final8.err; show gpurun_out/bench_r1_final8.json
rm -rf gpurun_out/prof_final5
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_final5 -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/prof_final5.log 2>&1
show gpurun_out/prof_final5.log
find gpurun_out/prof_final5 -name "*kernel_trace.csv" -delete
for wl in allcnnc resnet50; do
timeout 900 python bench.py --workload $wl --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/final_$wl.log 2>&1; show gpurun_out/final_$wl.log
done

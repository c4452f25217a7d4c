#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
show() { grep -a "\[bench\]" $1; tail -1 $1 | python -c "
import sys,json
try:
    d=json.loads(sys.stdin.read()); print(round(d['value'],1), d['config']['workload'][:30], '|', d['config']['matvec'][-110:], round(d['roofline']['frac'],3))
except Exception as e: print('no json', e)"; }
timeout 1800 python -m pytest tests/ -m gpu -q -x 2>&1 | tail -3
for i in 1 2 3 4; do
timeout 900 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r4_$i.log 2>&1; show gpurun_out/r4_$i.log
done
for wl in allcnnc resnet50; do
timeout 900 python bench.py --workload $wl --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/other_${wl}_1.log 2>&1; show gpurun_out/other_${wl}_1.log
done
for i in 1 2; do
rm -rf gpurun_out/prof_t$i
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_t$i -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/prof_t$i.log 2>&1
show gpurun_out/prof_t$i.log
find gpurun_out/prof_t$i -name "*kernel_trace.csv" -delete
done
rm -rf gpurun_out/miopen_db_after; cp -r pytorchhessianfree_amd/miopen_db gpurun_out/miopen_db_after

#!/bin/bash
cd $GRAFT_REPO_ROOT
show() { grep -a "\[bench\]" $1; grep -a '^{"metric"' $1 | tail -1 | python -c "
import sys,json
try:
    d=json.loads(sys.stdin.read()); print(round(d['value'],1), d['config']['workload'][:30], '|', d['config']['matvec'][-110:], round(d['roofline']['frac'],3))
except Exception as e: print('no json', e)"; }
timeout 900 python -m pytest tests/test_optimizer_gpu.py -q -x -k "conv_nets or resnet18" 2>&1 | tail -2
for i in 1 2 3; do
timeout 900 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r13_$i.log 2>&1; show gpurun_out/r13_$i.log
done

#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
show() { grep -a "\[bench\]" $1; grep -a '^{"metric"' $1 | tail -1 | python -c "
import sys,json
try:
    d=json.loads(sys.stdin.read()); print(round(d['value'],1), d['config']['workload'][:30], '|', d['config']['matvec'][-110:], round(d['roofline']['frac'],3))
except Exception as e: print('no json', e)"; }
timeout 900 python -m pytest tests/test_optimizer_gpu.py -q -x -k "slices or scatter or conv_nets or batchnorm or resnet18_step or channels_last" 2>&1 | tail -3
for cl in 1 0; do
timeout 900 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --channels-last $cl > gpurun_out/r7_$cl.log 2>&1; show gpurun_out/r7_$cl.log
done
timeout 900 python bench.py --workload resnet50 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r7_resnet50.log 2>&1; show gpurun_out/r7_resnet50.log

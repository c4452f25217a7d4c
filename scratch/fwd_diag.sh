#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 600 python3 scratch/fwd_diag.py resnet50 2>&1 | grep -a RESULT | cut -c1-900
show() { grep -a "\[bench\]" $1; tail -1 $1 | python -c "
import sys,json
try:
    d=json.loads(sys.stdin.read()); print(round(d['value'],1), d['config']['workload'][:40], '|', d['config']['matvec'][-70:], round(d['roofline']['frac'],3))
except Exception as e: print('no json', e)"; }
for wl in allcnnc resnet50; do for cl in 1 0; do
timeout 900 python bench.py --workload $wl --channels-last $cl --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/other_${wl}_$cl.log 2>&1; show gpurun_out/other_${wl}_$cl.log
done; done

#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
show() { grep -a "\[bench\]" $1; grep -a '^{"metric"' $1 | tail -1 | python -c "
import sys,json
try:
    d=json.loads(sys.stdin.read()); print(round(d['value'],1), d['config']['workload'][:30], '|', d['config']['matvec'][-110:], round(d['roofline']['frac'],3))
except Exception as e: print('no json', e)"; }
timeout 900 python -m pytest tests/test_optimizer_gpu.py -q -x -k "closed_form or step_trace or conv_nets" 2>&1 | tail -3
for i in 1 2; do
timeout 900 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r6_$i.log 2>&1; show gpurun_out/r6_$i.log
done
timeout 900 python bench.py --workload allcnnc --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r6_allcnnc.log 2>&1; show gpurun_out/r6_allcnnc.log

#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
show() { grep -a "\[bench\]" $1; tail -1 $1 | python -c "
import sys,json
try:
    d=json.loads(sys.stdin.read()); print(round(d['value'],1), d['config']['workload'][:30], '|', d['config']['matvec'][-110:], round(d['roofline']['frac'],3))
except Exception as e: print('no json', e)"; }
timeout 1800 python -m pytest tests/ -m gpu -q -x 2>&1 | tail -3
for wl in resnet18 allcnnc resnet50; do for cl in 1 0; do
timeout 900 python bench.py --workload $wl --channels-last $cl --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/other_${wl}_$cl.log 2>&1; show gpurun_out/other_${wl}_$cl.log
done; done
rm -rf gpurun_out/miopen_db_after; cp -r pytorchhessianfree_amd/miopen_db gpurun_out/miopen_db_after

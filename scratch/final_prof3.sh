#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
show() { grep -a "\[bench\]" $1; tail -1 $1 | python -c "
import sys,json
try:
    d=json.loads(sys.stdin.read()); print(round(d['value'],1), d['config']['workload'][:40], '|', d['config']['matvec'][-55:], round(d['roofline']['frac'],3))
except Exception as e: print('no json', e)"; }
timeout 1800 python -m pytest tests/ -m gpu -q 2>&1 | tail -3
timeout 600 python __graft_entry__.py smoke 2>&1 | tail -1
timeout 900 python bench.py > gpurun_out/bench_r1_final6.json 2> gpurun_out/bench_r1_final6.err; show gpurun_out/bench_r1_final6.json
rm -rf gpurun_out/prof_final3
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_final3 -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/prof_final3.log 2>&1
show gpurun_out/prof_final3.log
find gpurun_out/prof_final3 -name "*kernel_trace.csv" -delete
for wl in allcnnc resnet50; do for cl in 1 0; do
timeout 900 python bench.py --workload $wl --channels-last $cl --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/other_${wl}_$cl.log 2>&1; show gpurun_out/other_${wl}_$cl.log
done; done
rm -rf gpurun_out/miopen_db_after; cp -r pytorchhessianfree_amd/miopen_db gpurun_out/miopen_db_after

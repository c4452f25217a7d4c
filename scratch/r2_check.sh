#!/bin/bash
cd $GRAFT_REPO_ROOT
DB=$GRAFT_REPO_ROOT/gpurun_out/r2_db; rm -rf $DB; mkdir -p $DB; cp pytorchhessianfree_amd/miopen_db/*.txt $DB/
export MIOPEN_USER_DB_PATH=$DB
show() { grep -a "\[bench\]" $1; tail -1 $1 | python -c "
import sys,json
try:
    d=json.loads(sys.stdin.read()); print(round(d['value'],1), d['config']['matvec'][-60:], round(d['roofline']['frac'],3))
except Exception as e: print('no json', e)"; }
timeout 900 python -m pytest tests/test_optimizer_gpu.py -q -x -k "unpack or scatter or channels_last or conv_nets or resnet18 or batchnorm" 2>&1 | tail -5
for cl in 1 0; do
timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --channels-last $cl > gpurun_out/r2_bench_$cl.log 2>&1; show gpurun_out/r2_bench_$cl.log
done
echo "db lines $(cat $DB/*.ufdb.txt | wc -l)"

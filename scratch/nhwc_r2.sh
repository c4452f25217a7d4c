#!/bin/bash
# NHWC path with the column-per-block BatchNorm adjoint kernel: worker test, bench, kernel stats
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 300 python tests/gpu_workers/nhwc_small_net.py 2>&1 | grep -E "RESULT|Error|error" | cut -c1-300
DB=$GRAFT_REPO_ROOT/gpurun_out/cl_db; rm -rf $DB; mkdir -p $DB; cp pytorchhessianfree_amd/miopen_db/*.txt $DB/
export MIOPEN_USER_DB_PATH=$DB
for i in 1 2; do
timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --channels-last 1 > gpurun_out/bench_cl_$i.log 2>&1
grep "\[bench\]" gpurun_out/bench_cl_$i.log; tail -1 gpurun_out/bench_cl_$i.log | python -c "
import sys,json
try:
    d=json.loads(sys.stdin.read()); print(d['value'], d['config']['matvec'], d['roofline']['frac'])
except Exception as e: print('no json', e)"
done
rm -rf gpurun_out/prof_cl
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_cl -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --channels-last 1 > gpurun_out/prof_cl.log 2>&1
tail -1 gpurun_out/prof_cl.log | cut -c1-200
find gpurun_out/prof_cl -name "*kernel_trace.csv" -delete

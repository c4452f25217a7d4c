#!/bin/bash
# NHWC find-db records: one find-mode run (result checked), then immediate-mode runs on them
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
DB=$GRAFT_REPO_ROOT/gpurun_out/gen_db; rm -rf $DB; mkdir -p $DB; cp pytorchhessianfree_amd/miopen_db/*.txt $DB/
export MIOPEN_USER_DB_PATH=$DB
L() { cat $DB/*.ufdb.txt | wc -l; }
show() { grep -a "\[bench\]" $1; tail -1 $1 | python -c "
import sys,json
try:
    d=json.loads(sys.stdin.read()); print(round(d['value'],1), d['config']['matvec'][-60:], round(d['roofline']['frac'],3))
except Exception as e: print('no json', e)"; }
echo "db lines $(L)"
HF_NHWC_FIND=1 timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --channels-last 1 > gpurun_out/gen_find.log 2>&1; show gpurun_out/gen_find.log
HF_NHWC_FIND=1 timeout 300 python tests/gpu_workers/nhwc_small_net.py 2>&1 | grep -E "RESULT" | cut -c1-300
echo "db lines after find runs $(L)"
rm -rf gpurun_out/gen_db_snapshot; cp -r $DB gpurun_out/gen_db_snapshot
for i in 1 2 3; do
timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --channels-last 1 > gpurun_out/gen_imm_$i.log 2>&1; show gpurun_out/gen_imm_$i.log
done
timeout 300 python tests/gpu_workers/nhwc_small_net.py 2>&1 | grep -E "RESULT" | cut -c1-300
echo "db lines after immediate runs $(L)"
rm -rf gpurun_out/prof_cl3
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_cl3 -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --channels-last 1 > gpurun_out/prof_cl3.log 2>&1
show gpurun_out/prof_cl3.log
find gpurun_out/prof_cl3 -name "*kernel_trace.csv" -delete
echo "db lines at end $(L)"

#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
show() { grep -a "\[bench\]" $1; grep -a '^{"metric"' $1 | tail -1 | python -c "
import sys,json
try:
    d=json.loads(sys.stdin.read()); print(round(d['value'],1), d['config']['workload'][:30], '|', d['config']['matvec'][-110:], round(d['roofline']['frac'],3))
except Exception as e: print('no json', e)"; }
# NHWC records for All-CNN-C: one find-mode run, result checked by bench.py itself
HF_NHWC_FIND=1 timeout 900 python bench.py --workload allcnnc --channels-last 1 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/gen_allcnnc.log 2>&1; show gpurun_out/gen_allcnnc.log
rm -rf gpurun_out/miopen_db_gen; cp -r pytorchhessianfree_amd/miopen_db gpurun_out/miopen_db_gen
for i in 1 2; do
timeout 900 python bench.py --workload allcnnc --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/imm_allcnnc_$i.log 2>&1; show gpurun_out/imm_allcnnc_$i.log
done
timeout 900 python bench.py --workload resnet50 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/imm_resnet50.log 2>&1; show gpurun_out/imm_resnet50.log
# round-1 final artefacts: default bench (with the CPU baseline) and the rocprofv3 summary of the same command
timeout 1200 python bench.py > gpurun_out/bench_r1_final7.json 2> gpurun_out/bench_r1_final7.err; show gpurun_out/bench_r1_final7.json
rm -rf gpurun_out/prof_final4
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_final4 -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/prof_final4.log 2>&1
show gpurun_out/prof_final4.log
find gpurun_out/prof_final4 -name "*kernel_trace.csv" -delete

#!/bin/bash
cd $GRAFT_REPO_ROOT
show() { grep -a "\[bench\]" $1; grep -a '^{"metric"' $1 | tail -1 | python -c "
import sys,json
try:
    d=json.loads(sys.stdin.read()); print(round(d['value'],1), d['config']['workload'][:30], '|', d['config']['matvec'][-110:], round(d['roofline']['frac'],3))
except Exception as e: print('no json', e)"; }
timeout 900 python -m pytest tests/test_optimizer_gpu.py -q -x -k "centre_tap or conv_nets or channels_last" 2>&1 | tail -2
timeout 900 python bench.py --workload allcnnc --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r12_allcnnc.log 2>&1; show gpurun_out/r12_allcnnc.log
HF_NHWC_FIND=1 timeout 900 python bench.py --workload resnet50 --channels-last 1 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r12_resnet50.log 2>&1; show gpurun_out/r12_resnet50.log
rm -rf gpurun_out/miopen_db_r12; cp -r pytorchhessianfree_amd/miopen_db gpurun_out/miopen_db_r12
timeout 900 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r12_resnet18.log 2>&1; show gpurun_out/r12_resnet18.log

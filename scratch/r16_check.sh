#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 300 python tests/gpu_workers/nhwc_small_net.py 2>&1 | grep -E "RESULT|Error|error" | cut -c1-300
timeout 600 python -m pytest tests/test_optimizer_gpu.py -q -x -k "batchnorm or channels_last or slices or conv_nets or scatter" 2>&1 | tail -2
for i in 1 2; do
timeout 900 python bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | grep -a '^{"metric"' | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print(round(d['value'],1), d['config']['matvec'][-100:])"
done
HF_BN_ROWS_OFF=1 timeout 900 python bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | grep -a '^{"metric"' | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('rows kernel off:', round(d['value'],1))"

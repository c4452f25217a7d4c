#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_optimizer_gpu.py -q -x -k "refuses or centre_tap" 2>&1 | tail -12
timeout 900 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --force-dist 1 2>&1 | grep -a '^{"metric"' | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print(round(d['value'],1), d['config']['parallelism'], '|', d['config']['matvec'][-60:])"
HF_RCCL_DIRECT=1 timeout 900 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --force-dist 1 2>&1 | grep -a '^{"metric"' | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print(round(d['value'],1), 'direct rccl')"

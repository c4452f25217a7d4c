#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_optimizer_gpu.py -q -x -k "batchnorm or channels_last or slices or conv_nets" 2>&1 | tail -2
for cl in 1 0; do
timeout 900 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --channels-last $cl 2>/dev/null | grep -a '^{"metric"' | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print(round(d['value'],1), d['config']['matvec'][-100:])"
done

#!/bin/bash
cd $GRAFT_REPO_ROOT
export MIOPEN_USER_DB_PATH=$GRAFT_REPO_ROOT/gpurun_out/miopen_db6
mkdir -p $MIOPEN_USER_DB_PATH; cp profiles/miopen_db/* $MIOPEN_USER_DB_PATH/
SECONDS=0; python bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-200; echo "took $SECONDS"
SECONDS=0; python bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-200; echo "took $SECONDS"
python bench.py --workload allcnnc --steps 1 --warmup 1 --no-cpu-baseline --iters 30 2>&1 | tail -1 | cut -c1-200
timeout 900 python scratch/resnet_parity.py 1e-3 40 2>&1 | grep -v Warn | tail -16

#!/bin/bash
cd $GRAFT_REPO_ROOT
export MIOPEN_USER_DB_PATH=$GRAFT_REPO_ROOT/gpurun_out/miopen_db
mkdir -p $MIOPEN_USER_DB_PATH
echo "== first run (cold db)"; SECONDS=0; python bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>&1 | tail -2 | cut -c1-400
ls -la $MIOPEN_USER_DB_PATH
echo "first took $SECONDS s"; SECONDS=0; echo "== second run (warm db)"; python bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>&1 | tail -2 | cut -c1-400
echo "second took $SECONDS s"; SECONDS=0; echo "== allcnnc"; python bench.py --workload allcnnc --steps 1 --warmup 1 --no-cpu-baseline 2>&1 | tail -2 | cut -c1-600
ls -la $MIOPEN_USER_DB_PATH

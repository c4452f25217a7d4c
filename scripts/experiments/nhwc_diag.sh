#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for i in 1 2 3 4 5 6 7 8; do
rm -rf gpurun_out/prof_d
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_d -- python3 scripts/experiments/nhwc_diag.py stock_first 1 2>&1 | grep -a RESULT | cut -c1-600
done
for i in 1 2 3; do
rm -rf gpurun_out/prof_d
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_d -- python3 scripts/experiments/nhwc_diag.py op_first 1 2>&1 | grep -a RESULT | cut -c1-600
done
for i in 1 2 3; do
rm -rf gpurun_out/prof_d
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_d -- python3 scripts/experiments/nhwc_diag.py stock_first 0 2>&1 | grep -a RESULT | cut -c1-600
done
rm -rf gpurun_out/prof_d
# and the find step on NHWC shapes in a cold database, no profiler: still flaky without the CK wrw solver?
for i in 1 2 3 4 5 6; do
DB=$GRAFT_REPO_ROOT/gpurun_out/cold_$i; rm -rf $DB; mkdir -p $DB
MIOPEN_USER_DB_PATH=$DB HF_NHWC_FIND=1 timeout 300 python3 scripts/experiments/nhwc_diag.py stock_first 1 2>&1 | grep -a RESULT | cut -c1-300
done

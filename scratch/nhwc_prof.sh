#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
DB=$GRAFT_REPO_ROOT/gpurun_out/mode_db2; rm -rf $DB; mkdir -p $DB; cp pytorchhessianfree_amd/miopen_db/*.txt $DB/
export MIOPEN_USER_DB_PATH=$DB
timeout 200 python scratch/nhwc_mode.py 1 2>&1 | grep RESULT
rm -rf gpurun_out/prof_cl2
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_cl2 -- python3 scratch/nhwc_mode.py 0 > gpurun_out/prof_cl2.log 2>&1
grep RESULT gpurun_out/prof_cl2.log
find gpurun_out/prof_cl2 -name "*kernel_trace.csv" -delete

#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof4
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof4 -- python3 scratch/matvec_variants.py bn > gpurun_out/prof4.log 2>&1
tail -1 gpurun_out/prof4.log
find gpurun_out/prof4 -name "*kernel_trace.csv" -delete

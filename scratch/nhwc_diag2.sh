#!/bin/bash
cd $GRAFT_REPO_ROOT
for wl in allcnnc resnet50; do
timeout 600 python3 scratch/nhwc_diag.py stock_first 1 $wl 2>&1 | grep -a RESULT | cut -c1-1500
done
rm -rf gpurun_out/miopen_db_after; cp -r pytorchhessianfree_amd/miopen_db gpurun_out/miopen_db_after

#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python3 scratch/conv_scan.py resnet50 2>&1 | grep -a RESULT | cut -c1-250
rm -rf gpurun_out/miopen_db_scan; cp -r pytorchhessianfree_amd/miopen_db gpurun_out/miopen_db_scan

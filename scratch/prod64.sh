#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 300 python3 scratch/memset_repro.py 2>&1 | grep -a RESULT
timeout 900 python3 scratch/prod64_diag.py resnet50 2>&1 | grep -a -E "RESULT|fp64" | cut -c1-400
timeout 900 python3 scratch/prod64_diag.py resnet18 2>&1 | grep -a -E "RESULT|fp64" | cut -c1-400

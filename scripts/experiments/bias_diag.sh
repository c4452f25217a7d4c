#!/bin/bash
cd $GRAFT_REPO_ROOT
for mode in sum contig kernel clone; do
echo "mode $mode: $(HF_BIAS_MODE=$mode timeout 600 python3 scripts/experiments/nhwc_diag.py stock_first 1 allcnnc 2>&1 | grep -a RESULT | cut -c1-500)"
done

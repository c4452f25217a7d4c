#!/bin/bash
cd $GRAFT_REPO_ROOT
DB=$GRAFT_REPO_ROOT/gpurun_out/mode_db; rm -rf $DB; mkdir -p $DB; cp pytorchhessianfree_amd/miopen_db/*.txt $DB/
export MIOPEN_USER_DB_PATH=$DB
L() { cat $DB/*.ufdb.txt | wc -l; }
echo "db lines $(L)"
for i in 1 2 3; do timeout 200 python scratch/nhwc_mode.py 1 2>&1 | grep RESULT; echo "   db lines $(L)"; done
echo "--- immediate mode, warm db"
for i in 1 2 3 4 5 6 7 8 9 10 11 12; do timeout 200 python scratch/nhwc_mode.py 0 2>&1 | grep -E "RESULT|Error" | head -2; done
echo "   db lines $(L)"
echo "--- find mode, warm db"
for i in 1 2 3 4 5 6 7 8 9 10 11 12; do timeout 200 python scratch/nhwc_mode.py 1 2>&1 | grep -E "RESULT|Error" | head -2; done
echo "   db lines $(L)"
echo "--- NCHW immediate mode, warm db"
for i in 1 2; do timeout 200 python scratch/nhwc_mode.py 0 0 2>&1 | grep -E "RESULT|Error" | head -2; done

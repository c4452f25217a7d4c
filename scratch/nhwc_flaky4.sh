#!/bin/bash
cd $GRAFT_REPO_ROOT
for i in 0 1 2 3 4 5 6 7 8 9 10 11; do
  export MIOPEN_USER_DB_PATH=$PWD/gpurun_out/fl4_db_$i; rm -rf $MIOPEN_USER_DB_PATH; mkdir -p $MIOPEN_USER_DB_PATH
  if [ $((i % 2)) -eq 0 ]; then o=graph_first; else o=eager_first; fi
  echo "fresh db $i: $(timeout 300 python scratch/nhwc_flaky4.py $o 2>&1 | grep -E 'ORDER|Error' | tail -1)"
done

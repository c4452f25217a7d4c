#!/bin/bash
cd $GRAFT_REPO_ROOT
for i in 0 1 2 3 4 5 6 7 8 9; do
  export MIOPEN_USER_DB_PATH=$PWD/gpurun_out/fl3_db_$i; rm -rf $MIOPEN_USER_DB_PATH; mkdir -p $MIOPEN_USER_DB_PATH
  out=$(timeout 300 python scratch/nhwc_flaky2.py 0 1 2>&1 | grep trial | head -2 | cut -c1-90 | tr '\n' ';')
  echo "fresh db $i: $out"
done

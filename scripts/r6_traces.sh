#!/bin/bash
# Round-6 per-launch tables of ONE product (rocprofv3 kernel trace joined with the driver's launch log): the ResNet-18
# workload with train-mode BatchNorm on the final binary, and with stem + layer1 frozen.
OUT=${1:-gpurun_out/r6t}; mkdir -p $OUT; export TMPDIR=/tmp
for spec in "train --bn_train" "frozen --freeze_stem+layer1" "eval"; do
  set -- $spec; tag=$1; arg=$(echo "${2:-}" | sed 's/_/ /')
  rocprofv3 --kernel-trace --output-format csv -d /tmp/tr_$tag -- python3 scripts/engine_product_driver.py --products 10 $arg --out $OUT/launches_$tag.json > $OUT/trace_$tag.log 2>&1
  python3 scripts/product_trace_table.py $OUT/launches_$tag.json /tmp/tr_$tag > $OUT/r06_r18_${tag}_one_product_trace.txt 2>> $OUT/err.log
  tail -2 $OUT/r06_r18_${tag}_one_product_trace.txt
  rm -rf /tmp/tr_$tag
done

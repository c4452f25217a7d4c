#!/bin/bash
# Round-4 GPU batch 9: train-mode BatchNorm tangent / adjoint as ONE launch around a grid barrier
# (hf_bn_rows_train_apply) -- parity first (under a short timeout: the launch waits inside itself), then A/B.
O=gpurun_out/r4j; mkdir -p $O
timeout 600 python -m pytest tests/test_engine_gpu.py tests/test_session_gpu.py -q -m gpu -k "train_mode or folded" -x > $O/tests.log 2>&1; echo "tests rc=$?" >> $O/rc.log
if grep -q "tests rc=0" $O/rc.log; then
  for fuse in 1 0 1 0; do
    echo "== HF_BN_FUSE=$fuse" >> $O/train.jsonl
    HF_BN_FUSE=$fuse timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --bn train --no-beyond-l3 >> $O/train.jsonl 2>> $O/train.err
  done
  for rbk in 32 128; do
    echo "== HF_BN_FUSE=1 HF_BN_ROW_BLOCKS=$rbk" >> $O/train.jsonl
    HF_BN_ROW_BLOCKS=$rbk timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --bn train --no-beyond-l3 --no-step-timing >> $O/train.jsonl 2>> $O/train.err
  done
  rocprofv3 --kernel-trace --output-format csv -d $O/tr_train -- python3 scripts/engine_product_driver.py --workload resnet18 --products 6 --out $O/launches_train.json --bn train > $O/tr_train.log 2>&1
  python3 scripts/product_trace_table.py $O/launches_train.json $O/tr_train > $O/resnet18_train_one_product_trace.txt 2>> $O/tr_train.log
  rm -rf $O/tr_train
fi
cat $O/rc.log

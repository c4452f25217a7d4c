#!/bin/bash
# Round-4 GPU batch 12: train-mode forward with the statistics' finalisation in the normalising launch's prologue
# (hf_bn_forward_train) -- parity, then step_ms A/B on one box.
O=gpurun_out/r4m; mkdir -p $O
timeout 900 python -m pytest tests/test_engine_gpu.py tests/test_session_gpu.py -q -m gpu -k "train_mode or folded" -x > $O/tests.log 2>&1; echo "tests rc=$?" >> $O/rc.log
if grep -q "tests rc=0" $O/rc.log; then
  for rep in 1 2; do
    for pro in 1 0; do
      echo "== HF_BN_FWD_PROLOGUE=$pro" >> $O/train_fwd.jsonl
      HF_BN_FWD_PROLOGUE=$pro timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --bn train --no-beyond-l3 >> $O/train_fwd.jsonl 2>> $O/train.err
    done
  done
fi
cat $O/rc.log
rocprofv3 --kernel-trace --output-format csv -d $O/tr_train -- python3 scripts/engine_product_driver.py --workload resnet18 --products 6 --out $O/launches_train.json --bn train > $O/tr_train.log 2>&1
python3 scripts/product_trace_table.py $O/launches_train.json $O/tr_train > $O/resnet18_train_one_product_trace.txt 2>> $O/tr_train.log
rm -rf $O/tr_train

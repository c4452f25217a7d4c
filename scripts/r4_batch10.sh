#!/bin/bash
# Round-4 GPU batch 10: the four forms of the train-mode BatchNorm finalisation (HF_BN_TRAIN_FORM), same box.
O=gpurun_out/r4k; mkdir -p $O
timeout 900 python -m pytest tests/test_engine_gpu.py tests/test_session_gpu.py -q -m gpu -k "train_mode or folded" -x > $O/tests.log 2>&1; echo "tests rc=$?" >> $O/rc.log
if grep -q "tests rc=0" $O/rc.log; then
  for rep in 1 2; do
    for form in prologue barrier tail separate; do
      echo "== HF_BN_TRAIN_FORM=$form" >> $O/train_forms.jsonl
      HF_BN_TRAIN_FORM=$form timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --bn train --no-beyond-l3 --no-step-timing >> $O/train_forms.jsonl 2>> $O/train.err
    done
  done
  timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --bn train --no-beyond-l3 > $O/bench_train.json 2>> $O/train.err
  rocprofv3 --kernel-trace --output-format csv -d $O/tr_train -- python3 scripts/engine_product_driver.py --workload resnet18 --products 6 --out $O/launches_train.json --bn train > $O/tr_train.log 2>&1
  python3 scripts/product_trace_table.py $O/launches_train.json $O/tr_train > $O/resnet18_train_one_product_trace.txt 2>> $O/tr_train.log
  rm -rf $O/tr_train
fi
cat $O/rc.log

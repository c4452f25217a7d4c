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

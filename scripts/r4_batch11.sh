#!/bin/bash
# Round-4 GPU batch 11: the tangent's train-mode BatchNorm partial sums from the convolution's own epilogue
# (hf_conv2d_nhwc_group_slabs_bnsum) -- parity, then A/B on one box.
O=gpurun_out/r4l; mkdir -p $O
timeout 900 python -m pytest tests/test_engine_gpu.py tests/test_session_gpu.py tests/test_conv_gpu.py -q -m gpu -k "train_mode or folded or three_directions or grouped" -x > $O/tests.log 2>&1; echo "tests rc=$?" >> $O/rc.log
if grep -q "tests rc=0" $O/rc.log; then
  for rep in 1 2; do
    for epi in "1 256" "0 256" "1 128" "1 1024"; do
      set -- $epi
      echo "== HF_BN_EPILOGUE=$1 HF_BN_EPILOGUE_ROWS=$2" >> $O/train_epilogue.jsonl
      HF_BN_EPILOGUE=$1 HF_BN_EPILOGUE_ROWS=$2 timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --bn train --no-beyond-l3 --no-step-timing >> $O/train_epilogue.jsonl 2>> $O/train.err
    done
  done
  timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --bn train --no-beyond-l3 > $O/bench_train.json 2>> $O/train.err
  timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-beyond-l3 > $O/bench_eval.json 2>> $O/train.err
  rocprofv3 --kernel-trace --output-format csv -d $O/tr_train -- python3 scripts/engine_product_driver.py --workload resnet18 --products 6 --out $O/launches_train.json --bn train > $O/tr_train.log 2>&1
  python3 scripts/product_trace_table.py $O/launches_train.json $O/tr_train > $O/resnet18_train_one_product_trace.txt 2>> $O/tr_train.log
  rm -rf $O/tr_train
fi
cat $O/rc.log

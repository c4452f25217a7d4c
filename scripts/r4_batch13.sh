#!/bin/bash
# Round-4 GPU batch 13: the eval-mode BatchNorm tangent as straight-line code (k_chan_affine_v4s) -- parity, then A/B.
O=gpurun_out/r4n; mkdir -p $O
timeout 1200 python -m pytest tests/test_engine_gpu.py -q -m gpu -x -k "not diag_ef and not hessian_step" > $O/tests.log 2>&1; echo "tests rc=$?" >> $O/rc.log; timeout 600 python -m pytest tests/test_optimizer_gpu.py -q -m gpu -x -k "chan_affine or fuse" >> $O/tests.log 2>&1; echo "tests2 rc=$?" >> $O/rc.log
if grep -q "tests rc=0" $O/rc.log && grep -q "tests2 rc=0" $O/rc.log; then
  for rep in 1 2; do
    for st in 1 0; do
      echo "== HF_AFFINE_STRAIGHT=$st" >> $O/affine.jsonl
      HF_AFFINE_STRAIGHT=$st timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-beyond-l3 --no-step-timing >> $O/affine.jsonl 2>> $O/err.log
    done
  done
  for st in 1 0; do
    echo "== HF_AFFINE_STRAIGHT=$st allcnnc / resnet50" >> $O/affine.jsonl
    HF_AFFINE_STRAIGHT=$st timeout 600 python bench.py --workload allcnnc --steps 3 --warmup 1 --no-cpu-baseline --no-beyond-l3 --no-step-timing >> $O/affine.jsonl 2>> $O/err.log
    HF_AFFINE_STRAIGHT=$st timeout 600 python bench.py --workload resnet50 --steps 2 --warmup 1 --no-cpu-baseline --no-beyond-l3 --no-step-timing >> $O/affine.jsonl 2>> $O/err.log
  done
fi
cat $O/rc.log

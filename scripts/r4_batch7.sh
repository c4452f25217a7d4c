#!/bin/bash
# Round-4 GPU batch 7: Hessian extras with ONE fork / ONE join (HF_HESSIAN_PARALLEL=2) against the per-unit form (1).
O=gpurun_out/r4h; mkdir -p $O
run() { name=$1; shift; "$@" > $O/$name.log 2>&1; echo "$name rc=$?" >> $O/rc.log; }
HF_HESSIAN_PARALLEL=2 python -m pytest tests/test_engine_gpu.py tests/test_optimizer_gpu.py tests/test_session_gpu.py tests/test_acc_session_gpu.py -q -m gpu -k "hessian or config4" > $O/tests_mode2.log 2>&1; echo "tests_mode2 rc=$?" >> $O/rc.log
: > $O/hessian_modes.jsonl
for rep in 1 2; do
  for par in 1 2; do
    echo "== HF_HESSIAN_PARALLEL=$par resnet18 rep $rep" >> $O/hessian_modes.jsonl
    HF_HESSIAN_PARALLEL=$par python bench.py --workload resnet18 --curvature hessian --steps 3 --warmup 1 --no-cpu-baseline --no-beyond-l3 --no-step-timing >> $O/hessian_modes.jsonl 2>> $O/hessian.err
    echo "== HF_HESSIAN_PARALLEL=$par config4 rep $rep" >> $O/hessian_modes.jsonl
    HF_HESSIAN_PARALLEL=$par python bench.py --workload allcnnc --curvature hessian --precond 1 --damping 1.0 --steps 3 --warmup 1 --no-cpu-baseline --no-beyond-l3 --no-step-timing >> $O/hessian_modes.jsonl 2>> $O/hessian.err
  done
done
run bottleneck python -m pytest tests/test_session_gpu.py -q -m gpu -k "bottleneck"
cat $O/rc.log

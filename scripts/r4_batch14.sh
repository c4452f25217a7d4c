#!/bin/bash
# Round-4 GPU batch 14: BatchNorm adjoint rows kernel with its first slab batch straight-line behind the row loads --
# parity, then A/B against the library built from the previous source (build_variants/libhfpcg_rows_loop.so).
O=gpurun_out/r4o; mkdir -p $O
timeout 1200 python -m pytest tests/test_engine_gpu.py -q -m gpu -x -k "not diag_ef and not hessian_step" > $O/tests.log 2>&1; echo "tests rc=$?" >> $O/rc.log
timeout 900 python -m pytest tests/test_session_gpu.py tests/test_optimizer_gpu.py -q -m gpu -x -k "train_mode or chan_affine or fuse or resnet18_default or session_steps" >> $O/tests.log 2>&1; echo "tests2 rc=$?" >> $O/rc.log
if grep -q "tests rc=0" $O/rc.log && grep -q "tests2 rc=0" $O/rc.log; then
  OLD=$GRAFT_REPO_ROOT/build_variants/libhfpcg_rows_loop.so
  for rep in 1 2; do
    echo "== new" >> $O/rows.jsonl
    timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-beyond-l3 --no-step-timing >> $O/rows.jsonl 2>> $O/err.log
    echo "== old (loop)" >> $O/rows.jsonl
    HF_PCG_LIB=$OLD timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-beyond-l3 --no-step-timing >> $O/rows.jsonl 2>> $O/err.log
  done
  for args in "--bn train" "--curvature hessian" "--workload allcnnc" "--workload resnet50"; do
    echo "== new $args" >> $O/rows.jsonl
    timeout 600 python bench.py $args --steps 3 --warmup 1 --no-cpu-baseline --no-beyond-l3 --no-step-timing >> $O/rows.jsonl 2>> $O/err.log
    echo "== old (loop) $args" >> $O/rows.jsonl
    HF_PCG_LIB=$OLD timeout 600 python bench.py $args --steps 3 --warmup 1 --no-cpu-baseline --no-beyond-l3 --no-step-timing >> $O/rows.jsonl 2>> $O/err.log
  done
fi
cat $O/rc.log

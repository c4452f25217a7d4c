#!/bin/bash
# Round-5 A/B on one box: strided data gradients by residue class unsplit (default) against split like every other 128-wide
# launch (build_variants/libhfpcg_dsplit.so, -DHF_DCLASS_NOSPLIT=0).
OUT=${1:-gpurun_out/dclass}; mkdir -p $OUT
timeout 600 python -m pytest tests/test_conv_gpu.py -q -m gpu -x 2>&1 | tail -2 > $OUT/conv_tests.log
for lib in "" "$PWD/build_variants/libhfpcg_dsplit.so"; do
  tag=$([ -z "$lib" ] && echo nosplit || echo split)
  HF_PCG_LIB=$lib python scripts/conv_kernel_bench.py --big 1 --no-reduce 1 > $OUT/conv_kernel_bench_$tag.jsonl 2>> $OUT/err.log
done
: > $OUT/dclass_ab.jsonl
for rep in 1 2; do
  for lib in "" "$PWD/build_variants/libhfpcg_dsplit.so"; do
    for args in "--workload allcnnc --curvature hessian --precond 1 --damping 1.0" "--workload allcnnc" "--workload resnet50"; do
      echo "== HF_PCG_LIB=$lib $args" >> $OUT/dclass_ab.jsonl
      HF_PCG_LIB=$lib python bench.py $args --steps 3 --warmup 1 --no-cpu-baseline --no-step-timing >> $OUT/dclass_ab.jsonl 2>> $OUT/err.log
    done
  done
done
cat $OUT/conv_tests.log

#!/bin/bash
# Round-5 A/B on one box: workgroups a 128-wide convolution launch is split towards (768 default; 512; 1024), whole bench.
OUT=${1:-gpurun_out/bigblocks}; mkdir -p $OUT
: > $OUT/big_blocks_ab.jsonl
for rep in 1 2; do
  for lib in "" "$PWD/build_variants/libhfpcg_big512.so" "$PWD/build_variants/libhfpcg_big1024.so"; do
    for args in "--workload allcnnc --curvature hessian --precond 1 --damping 1.0" "--workload allcnnc"; do
      echo "== HF_PCG_LIB=$lib $args" >> $OUT/big_blocks_ab.jsonl
      HF_PCG_LIB=$lib python bench.py $args --steps 3 --warmup 1 --no-cpu-baseline --no-step-timing >> $OUT/big_blocks_ab.jsonl 2>> $OUT/err.log
    done
  done
done

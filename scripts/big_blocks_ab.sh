#!/bin/bash
# Round-5 A/B on one box: workgroups a 128-wide convolution launch is split towards (HF_BIG_TARGET_BLOCKS) and the work from
# which a problem takes a 128-wide configuration (HF_BIG_MIN_WORK), whole bench; variants under build_variants/.
OUT=${1:-gpurun_out/bigblocks}; mkdir -p $OUT
: > $OUT/big_blocks_ab2.jsonl
for rep in 1 2; do
  for lib in "" $PWD/build_variants/libhfpcg_big512_w6144.so $PWD/build_variants/libhfpcg_big512_w4096.so $PWD/build_variants/libhfpcg_big512_w3072.so $PWD/build_variants/libhfpcg_big384_w6144.so $PWD/build_variants/libhfpcg_big256_w6144.so; do
    for args in "--workload allcnnc --curvature hessian --precond 1 --damping 1.0" "--workload allcnnc"; do
      echo "== HF_PCG_LIB=$lib $args" >> $OUT/big_blocks_ab2.jsonl
      HF_PCG_LIB=$lib python bench.py $args --steps 3 --warmup 1 --no-cpu-baseline --no-step-timing >> $OUT/big_blocks_ab2.jsonl 2>> $OUT/err.log
    done
  done
done

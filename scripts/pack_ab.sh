#!/bin/bash
# Round-5 A/B on one box: the gather with 16-deep slab batches + finer chunks for many-slab sources (default) against the
# round-4 form (build_variants/libhfpcg_pack8.so, -DHF_PACK_DEEP=0).   bash scripts/pack_ab.sh <out dir>
set -u
OUT=${1:-gpurun_out/packab}
mkdir -p "$OUT"
: > "$OUT/pack_ab.jsonl"
for rep in 1 2; do
  for lib in "" "$PWD/build_variants/libhfpcg_pack8.so"; do
    for args in "--workload allcnnc --curvature hessian --precond 1 --damping 1.0" "--workload allcnnc" "--workload resnet50"; do
      echo "== HF_PCG_LIB=$lib $args" >> "$OUT/pack_ab.jsonl"
      HF_PCG_LIB=$lib python bench.py $args --steps 3 --warmup 1 --no-cpu-baseline --no-step-timing >> "$OUT/pack_ab.jsonl" 2>> "$OUT/err.log"
    done
  done
done

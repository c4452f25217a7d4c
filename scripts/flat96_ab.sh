#!/bin/bash
# Round-5 A/B on one box: weight gradients of 96 / 192-channel layers on the Flat96 configuration (96 x 128 tiles, columns
# flat over (tap, channel); default) against the build without it (build_variants/libhfpcg_noflat.so, -DHF_CONV_FLAT96=0).
set -u
OUT=${1:-gpurun_out/flatab}
mkdir -p "$OUT"
timeout 600 python -m pytest tests/test_conv_gpu.py -q -m gpu -x 2>&1 | tail -5 > "$OUT/conv_tests.log"
for lib in "" "$PWD/build_variants/libhfpcg_noflat.so"; do
  tag=$([ -z "$lib" ] && echo flat || echo noflat)
  HF_PCG_LIB=$lib python scripts/conv_kernel_bench.py --big 1 --no-reduce 1 > "$OUT/conv_kernel_bench_$tag.jsonl" 2>> "$OUT/err.log"
done
: > "$OUT/flat_ab.jsonl"
for rep in 1 2; do
  for lib in "" "$PWD/build_variants/libhfpcg_noflat.so"; do
    for args in "--workload allcnnc --curvature hessian --precond 1 --damping 1.0" "--workload allcnnc"; do
      echo "== HF_PCG_LIB=$lib $args" >> "$OUT/flat_ab.jsonl"
      HF_PCG_LIB=$lib python bench.py $args --steps 3 --warmup 1 --no-cpu-baseline --no-step-timing >> "$OUT/flat_ab.jsonl" 2>> "$OUT/err.log"
    done
  done
done

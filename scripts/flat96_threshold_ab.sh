OUT=gpurun_out/r5i; mkdir -p $OUT
timeout 600 python -m pytest tests/test_conv_gpu.py -q -m gpu -x 2>&1 | tail -2 > $OUT/conv_tests.log
python scripts/conv_kernel_bench.py --big 1 --no-reduce 1 > $OUT/conv_kernel_bench_flat2048.jsonl 2>> $OUT/err.log
: > $OUT/flat_ab2.jsonl
for rep in 1 2; do
  for lib in "" "$PWD/build_variants/libhfpcg_flat6144.so"; do
    for args in "--workload allcnnc --curvature hessian --precond 1 --damping 1.0" "--workload allcnnc"; do
      echo "== HF_PCG_LIB=$lib $args" >> $OUT/flat_ab2.jsonl
      HF_PCG_LIB=$lib python bench.py $args --steps 3 --warmup 1 --no-cpu-baseline --no-step-timing >> $OUT/flat_ab2.jsonl 2>> $OUT/err.log
    done
  done
done
cat $OUT/conv_tests.log

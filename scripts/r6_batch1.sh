#!/bin/bash
# Round-6 batch 1 (one box): full GPU suite, then K3 non-temporal A/B at large N, acc_step / frozen-layer benches.
OUT=${1:-gpurun_out/r6b}; mkdir -p $OUT
python -m pytest tests -m gpu -q -x --durations=15 -p no:cacheprovider > $OUT/suite.log 2>&1
tail -3 $OUT/suite.log
: > $OUT/pcg_kernel_bench_k3nt.jsonl
for rep in 1 2; do
  for lib in "" $PWD/build_variants/libhfpcg_k3nt_off.so; do
    echo "== HF_PCG_LIB=$lib" >> $OUT/pcg_kernel_bench_k3nt.jsonl
    HF_PCG_LIB=$lib python scripts/pcg_kernel_bench.py --sizes 11175370,25557032,67108864,100000000 >> $OUT/pcg_kernel_bench_k3nt.jsonl 2>> $OUT/err.log
  done
done
python scripts/pcg_kernel_bench.py --sizes 1387108,11175370,25557032 --precond 1 > $OUT/pcg_kernel_bench_precond.jsonl 2>> $OUT/err.log
python bench.py --acc 16,16 --steps 5 --warmup 2 --no-cpu-baseline --no-beyond-l3 --no-train-bn > $OUT/bench_acc_16_16.json 2>> $OUT/err.log
python bench.py --freeze stem+layer1 --steps 5 --warmup 2 --no-cpu-baseline --no-beyond-l3 --no-train-bn > $OUT/bench_frozen.json 2>> $OUT/err.log
python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-beyond-l3 --no-train-bn > $OUT/bench_plain.json 2>> $OUT/err.log
tail -c 600 $OUT/err.log

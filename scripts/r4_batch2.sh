#!/bin/bash
# Round-4 GPU batch 2: re-run of the tests fixed after batch 1, NT variants of K2 / K3, train-mode bench after the
# parallel finalisation, ResNet-50 one-product trace, and the FULL GPU suite with durations.
O=gpurun_out/r4d; mkdir -p $O
run() { name=$1; shift; "$@" > $O/$name.log 2>&1; echo "$name rc=$?" >> $O/rc.log; }
run fixed python -m pytest tests/test_engine_gpu.py tests/test_session_gpu.py tests/test_distributed_gpu.py -q -m gpu -k "hessian_step or folded or train_mode or bench_gpus_2"
for v in main k1plain k2nt k3nt k23nt; do
  lib=pytorchhessianfree_amd/csrc/variants/libhfpcg_$v.so
  [ $v = main ] && lib=pytorchhessianfree_amd/csrc/libhfpcg.so
  echo "== $v" >> $O/pcg_nt_variants.jsonl
  HF_PCG_LIB=$PWD/$lib python scripts/pcg_kernel_bench.py --sizes 11175370,100000000 >> $O/pcg_nt_variants.jsonl 2>> $O/pcg_nt_variants.err
done
for v in main k23nt; do
  lib=pytorchhessianfree_amd/csrc/variants/libhfpcg_$v.so
  [ $v = main ] && lib=pytorchhessianfree_amd/csrc/libhfpcg.so
  HF_PCG_LIB=$PWD/$lib python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-step-timing > $O/bench_n1_$v.json 2> $O/bench_n1_$v.err; echo "bench_n1_$v rc=$?" >> $O/rc.log
done
python bench.py --steps 3 --warmup 1 --no-cpu-baseline --bn train > $O/bench_train.json 2> $O/bench_train.err; echo "bench_train rc=$?" >> $O/rc.log
HF_BN_FOLD=0 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --bn train --no-step-timing > $O/bench_train_nofold.json 2> $O/bench_train_nofold.err; echo "bench_train_nofold rc=$?" >> $O/rc.log
python bench.py --steps 3 --warmup 1 --no-cpu-baseline --workload allcnnc --curvature hessian --precond 1 --damping 1.0 > $O/bench_config4_d1.json 2> $O/bench_config4_d1.err; echo "bench_config4_d1 rc=$?" >> $O/rc.log
python bench.py --steps 2 --warmup 1 --no-cpu-baseline --workload resnet50 > $O/bench_resnet50.json 2> $O/bench_resnet50.err; echo "bench_resnet50 rc=$?" >> $O/rc.log
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/tr_r50 -- python3 scripts/engine_product_driver.py --workload resnet50 --products 6 --out $O/launches_r50.json > $O/tr_r50.log 2>&1
python3 scripts/product_trace_table.py $O/launches_r50.json $O/tr_r50 > $O/resnet50_one_product_trace.txt 2>> $O/tr_r50.log
find $O/tr_r50 -name "*kernel_stats.csv" -exec cp {} $O/resnet50_product_kernel_stats.csv \;
find $O/tr_r50 -name "*kernel_trace.csv" -exec cp {} $O/resnet50_product_kernel_trace.csv \;
rm -rf $O/tr_r50
rocprofv3 --kernel-trace --output-format csv -d $O/tr_train -- python3 scripts/engine_product_driver.py --workload resnet18 --products 6 --out $O/launches_train.json --bn train > $O/tr_train.log 2>&1
python3 scripts/product_trace_table.py $O/launches_train.json $O/tr_train > $O/resnet18_train_one_product_trace.txt 2>> $O/tr_train.log
rm -rf $O/tr_train
( time python -m pytest tests -q -m gpu --durations=30 ) > $O/full_suite.log 2>&1; echo "full_suite rc=$?" >> $O/rc.log
cat $O/rc.log

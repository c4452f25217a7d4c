#!/bin/bash
# Round-4 GPU batch 6: weight gradients on the side branch (HF_ADJ_SPLIT) A/B on three workloads; acc Hessian test.
O=gpurun_out/r4g; mkdir -p $O
run() { name=$1; shift; "$@" > $O/$name.log 2>&1; echo "$name rc=$?" >> $O/rc.log; }
run tests python -m pytest tests/test_acc_session_gpu.py tests/test_engine_gpu.py tests/test_optimizer_gpu.py -q -m gpu -k "hessian or resnet18_engine_product or diag_ef or get_preconditioner or transposed or declines"
HF_ADJ_SPLIT=1 python -m pytest tests/test_engine_gpu.py tests/test_session_gpu.py -q -m gpu -k "resnet18_engine_product or allcnnc_plain or session_steps_match or bottleneck_net_engine" > $O/tests_split.log 2>&1; echo "tests_split rc=$?" >> $O/rc.log
: > $O/adj_split.jsonl
for rep in 1 2; do
  for sp in 0 1; do
    for wl in resnet18 resnet50 allcnnc; do
      echo "== HF_ADJ_SPLIT=$sp $wl rep $rep" >> $O/adj_split.jsonl
      HF_ADJ_SPLIT=$sp python bench.py --workload $wl --steps 3 --warmup 1 --no-cpu-baseline --no-step-timing --no-beyond-l3 >> $O/adj_split.jsonl 2>> $O/adj_split.err
    done
  done
done
export TMPDIR=/tmp
for spec in "allcnnc hessian ach" "allcnnc ggn acg"; do
  set -- $spec
  rocprofv3 --kernel-trace --output-format csv -d $O/tr_$3 -- python3 scripts/engine_product_driver.py --workload $1 --curvature $2 --products 8 --out $O/launches_$3.json > $O/tr_$3.log 2>&1
  find $O/tr_$3 -name "*kernel_trace.csv" -exec cp {} $O/$3_kernel_trace.csv \;
  rm -rf $O/tr_$3
done
python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-beyond-l3 --workload allcnnc --curvature hessian --precond 1 --damping 1.0 > $O/bench_config4.json 2> $O/bench_config4.err; echo "bench_config4 rc=$?" >> $O/rc.log
( time python -m pytest tests -q -m gpu --durations=12 ) > $O/full_suite.log 2>&1; echo "full_suite rc=$?" >> $O/rc.log
cat $O/rc.log

#!/bin/bash
# Round-4 final pass on the final binary: the whole GPU suite, the other-workloads leg, the train-mode trace.
O=gpurun_out/r4v; mkdir -p $O
export TMPDIR=/tmp
( time python -m pytest tests -q -m gpu --durations=10 ) > $O/full_suite.log 2>&1; echo "full_suite rc=$?" >> $O/rc.log
: > $O/r04_other_workloads.jsonl
for args in "--workload allcnnc" "--workload allcnnc --curvature hessian --precond 1 --damping 1.0" "--workload resnet50" "--workload resnet18 --bn train" "--workload resnet18 --curvature hessian" "--workload resnet18 --acc 16,16"; do
  python bench.py $args --steps 3 --warmup 1 >> $O/r04_other_workloads.jsonl 2>> $O/other.err
done
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/r04_bench_n1.json 2> $O/bench.err; echo "bench rc=$?" >> $O/rc.log
rocprofv3 --kernel-trace --output-format csv -d $O/tr_train -- python3 scripts/engine_product_driver.py --workload resnet18 --products 8 --out $O/launches_train.json --bn train > $O/tr_train.log 2>&1
python3 scripts/product_trace_table.py $O/launches_train.json $O/tr_train > $O/r04_r18_train_one_product_trace.txt 2>> $O/tr_train.log
rm -rf $O/tr_train
tail -4 $O/full_suite.log; cat $O/rc.log; tail -2 $O/r04_r18_train_one_product_trace.txt

#!/bin/bash
# Round-4 GPU batch 5: Hessian extras on a parallel branch, LDS-tiled transposed copies, K3 default policy.
O=gpurun_out/r4f; mkdir -p $O
export TMPDIR=/tmp
run() { name=$1; shift; "$@" > $O/$name.log 2>&1; echo "$name rc=$?" >> $O/rc.log; }
run tests python -m pytest tests/test_engine_gpu.py tests/test_optimizer_gpu.py tests/test_session_gpu.py tests/test_acc_session_gpu.py -q -m gpu -k "hessian or transposed or bottleneck or config4 or unpack"
: > $O/hessian.jsonl
for par in 1 0; do
  echo "== HF_HESSIAN_PARALLEL=$par resnet18" >> $O/hessian.jsonl
  HF_HESSIAN_PARALLEL=$par python bench.py --workload resnet18 --curvature hessian --steps 3 --warmup 1 --no-cpu-baseline --no-beyond-l3 >> $O/hessian.jsonl 2>> $O/hessian.err
  echo "== HF_HESSIAN_PARALLEL=$par config4" >> $O/hessian.jsonl
  HF_HESSIAN_PARALLEL=$par python bench.py --workload allcnnc --curvature hessian --precond 1 --damping 1.0 --steps 3 --warmup 1 --no-cpu-baseline --no-beyond-l3 >> $O/hessian.jsonl 2>> $O/hessian.err
done
: > $O/nt_ab.jsonl
for rep in 1 2; do
  for nt in 4000000 999999999999; do
    echo "== HF_PCG_NT_MIN=$nt rep $rep" >> $O/nt_ab.jsonl
    HF_PCG_NT_MIN=$nt python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-step-timing --no-beyond-l3 >> $O/nt_ab.jsonl 2>> $O/nt_ab.err
  done
done
rocprofv3 --kernel-trace --output-format csv -d $O/tr_h -- python3 scripts/engine_product_driver.py --workload resnet18 --curvature hessian --products 8 --out $O/launches_h.json > $O/tr_h.log 2>&1
python3 scripts/product_trace_table.py $O/launches_h.json $O/tr_h > $O/r04_r18_hessian_one_product_trace.txt 2>> $O/tr_h.log
find $O/tr_h -name "*kernel_trace.csv" -exec cp {} $O/r18_hessian_kernel_trace.csv \;
rm -rf $O/tr_h
cat $O/rc.log

#!/bin/bash
# Round-4 GPU batch 16: counter table and one-product traces again for the workloads whose products contain launches of
# other libraries (unfused classifier head): ResNet-50 topology, ResNet-18 Hessian.
O=gpurun_out/r4q2; mkdir -p $O
DRV_PRODUCTS=8 DRV_ARGS="--workload resnet50" bash scripts/run_engine_counters.sh $O/pmc_r50 > $O/pmc_r50.log 2>&1
cp $O/pmc_r50/engine_kernel_counters.json $O/r04_r50_engine_kernel_counters.json 2>/dev/null
cat $O/pmc_r50/table.err
rm -rf $O/pmc_r50/trace $O/pmc_r50/fetch $O/pmc_r50/write $O/pmc_r50/sq
for spec in "resnet18 hessian eval r18_hessian" "resnet50 ggn eval r50"; do
  set -- $spec
  rocprofv3 --kernel-trace --output-format csv -d "$O/tr_$4" -- python3 scripts/engine_product_driver.py --workload $1 --curvature $2 --bn $3 --products 8 --out "$O/launches_$4.json" > "$O/tr_$4.log" 2>&1
  python3 scripts/product_trace_table.py "$O/launches_$4.json" "$O/tr_$4" > "$O/r04_$4_one_product_trace.txt" 2>> "$O/tr_$4.log"
  tail -2 "$O/r04_$4_one_product_trace.txt"; tail -2 "$O/tr_$4.log"
  rm -rf "$O/tr_$4"
done

#!/bin/bash
# Round-4 GPU batch 15: PMC counter tables of one product for the ResNet-50 topology and for train-mode ResNet-18.
O=gpurun_out/r4p2; mkdir -p $O
DRV_PRODUCTS=8 DRV_ARGS="--workload resnet50" bash scripts/run_engine_counters.sh $O/pmc_r50 > $O/pmc_r50.log 2>&1
cp $O/pmc_r50/engine_kernel_counters.json $O/r04_r50_engine_kernel_counters.json 2>/dev/null
DRV_PRODUCTS=12 DRV_ARGS="--bn train" bash scripts/run_engine_counters.sh $O/pmc_train > $O/pmc_train.log 2>&1
cp $O/pmc_train/engine_kernel_counters.json $O/r04_r18_train_engine_kernel_counters.json 2>/dev/null
rm -rf $O/pmc_r50/trace $O/pmc_r50/fetch $O/pmc_r50/write $O/pmc_r50/sq $O/pmc_train/trace $O/pmc_train/fetch $O/pmc_train/write $O/pmc_train/sq
ls -la $O

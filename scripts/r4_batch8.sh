#!/bin/bash
# Round-4 GPU batch 8: split-K cap for weight-gradient problems with few output tiles (HF_CONV_FEW_TILES) A/B.
O=gpurun_out/r4i; mkdir -p $O
: > $O/few_tiles.jsonl
for ft in 2 4 8; do
  for wl in resnet50 resnet18 allcnnc; do
    echo "== HF_CONV_FEW_TILES=$ft $wl" >> $O/few_tiles.jsonl
    HF_CONV_FEW_TILES=$ft python bench.py --workload $wl --steps 3 --warmup 1 --no-cpu-baseline --no-step-timing --no-beyond-l3 >> $O/few_tiles.jsonl 2>> $O/few_tiles.err
  done
done
HF_CONV_FEW_TILES=4 python -m pytest tests/test_conv_gpu.py tests/test_engine_gpu.py -q -m gpu -k "three_directions or bottleneck_net_engine or resnet18_engine_product" > $O/tests.log 2>&1; echo "tests rc=$?" >> $O/rc.log
python bench.py --workload resnet18 --curvature hessian --steps 3 --warmup 1 --no-cpu-baseline --no-beyond-l3 > $O/r18_hessian.json 2>> $O/few_tiles.err
cat $O/rc.log

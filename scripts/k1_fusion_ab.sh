#!/bin/bash
# Round-5 A/B on one box: the PCG curvature scalar from the product's gather (hf_pack_ex_curv, default) against the K1
# launch (HF_FUSE_CURVATURE=0) -- bench line + rocprofv3 kernel statistics of each.   bash scripts/k1_fusion_ab.sh <out dir>
set -u
OUT=${1:-gpurun_out/k1ab}
mkdir -p "$OUT"
export TMPDIR=/tmp
ARGS="--steps 6 --warmup 2 --no-cpu-baseline --no-train-bn --no-step-timing --no-beyond-l3"
for rep in 1 2; do
  for m in 1 0; do
    HF_FUSE_CURVATURE=$m python bench.py $ARGS > "$OUT/bench_fuse${m}_rep$rep.json" 2>> "$OUT/err.log"
  done
done
for m in 1 0; do
  export HF_FUSE_CURVATURE=$m
  rocprofv3 --kernel-trace --stats --output-format csv -d "/tmp/prof_fuse$m" -- python3 bench.py $ARGS > "$OUT/bench_fuse${m}_under_rocprof.json" 2>> "$OUT/err.log"
  find "/tmp/prof_fuse$m" -name "*kernel_stats.csv" -exec cp {} "$OUT/kernel_stats_fuse$m.csv" \;
  rm -rf "/tmp/prof_fuse$m"
done

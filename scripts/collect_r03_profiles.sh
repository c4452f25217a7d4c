#!/bin/bash
# Round-3 evidence run (on the GPU box, from the repo root):  bash scripts/collect_r03_profiles.sh gpurun_out/r3p
set -u
OUT=${1:-gpurun_out/r3p}
mkdir -p "$OUT"
export TMPDIR=/tmp
# headline, as the driver runs it
python bench.py > "$OUT/r03_bench_n1.json" 2> "$OUT/r03_bench_n1.err"
# rocprofv3 summary of the same command (no CPU leg, no step leg: kernels of the timed loop)
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_bench" -- python3 bench.py --no-cpu-baseline --no-step-timing > "$OUT/r03_bench_n1_under_rocprof.json" 2> "$OUT/prof_bench.log"
find "$OUT/prof_bench" -name "*kernel_stats.csv" -exec cp {} "$OUT/r03_bench_n1_kernel_stats.csv" \;
rm -rf "$OUT/prof_bench"
# other workloads
: > "$OUT/r03_other_workloads.jsonl"
for args in "--workload allcnnc" "--workload allcnnc --curvature hessian --precond 1 --damping 1.0" "--workload resnet50" "--workload resnet18 --bn train"; do
  python bench.py $args --steps 2 >> "$OUT/r03_other_workloads.jsonl" 2>> "$OUT/other.err"
done
: > "$OUT/r03_autograd_paths.jsonl"
for args in "--workload allcnnc" "--workload allcnnc --curvature hessian --precond 1 --damping 1.0" "--workload resnet18" "--workload resnet50"; do
  HF_ENGINE=0 python bench.py $args --steps 2 --no-cpu-baseline --no-step-timing >> "$OUT/r03_autograd_paths.jsonl" 2>> "$OUT/other.err"
done
# data-parallel paths that one GPU can exercise
python bench.py --force-dist 1 --chunk 0 --no-cpu-baseline --no-step-timing > "$OUT/r03_bench_1rank_rccl.json" 2>> "$OUT/dp.err"
python bench.py --force-dist 1 --chunk 1 --no-cpu-baseline --no-step-timing > "$OUT/r03_bench_1rank_rccl_chunked.json" 2>> "$OUT/dp.err"
python bench.py --gpus 2 --steps 2 --no-cpu-baseline > "$OUT/r03_bench_2ranks_one_gpu_gloo.json" 2>> "$OUT/dp.err"
# PCG vector kernels at the three vector sizes
python scripts/pcg_kernel_bench.py > "$OUT/r03_pcg_kernel_bench.jsonl" 2>> "$OUT/other.err"
# complete steps: generic path vs persistent session
ONLY=engine python scripts/experiments/step_time.py > "$OUT/r03_step_time_session.txt" 2>&1
HF_SESSION=0 ONLY=engine python scripts/experiments/step_time.py > "$OUT/r03_step_time_generic.txt" 2>&1
python scripts/experiments/session_profile.py > "$OUT/r03_session_step_profile.txt" 2>&1
# convolution kernels on the large-map shapes
python scripts/conv_kernel_bench.py --big 1 --no-reduce 1 > "$OUT/r03_conv_kernel_bench_big.jsonl" 2>> "$OUT/other.err"
python scripts/conv_kernel_bench.py --no-reduce 1 > "$OUT/r03_conv_kernel_bench.jsonl" 2>> "$OUT/other.err"
# one-product traces
for cur in ggn hessian; do
  rocprofv3 --kernel-trace --output-format csv -d "$OUT/tr_$cur" -- python3 scripts/engine_product_driver.py --workload allcnnc --curvature $cur --products 12 --out "$OUT/launches_$cur.json" > "$OUT/tr_$cur.log" 2>&1
  python3 scripts/product_trace_table.py "$OUT/launches_$cur.json" "$OUT/tr_$cur" > "$OUT/r03_allcnnc_${cur}_one_product_trace.txt"
  rm -rf "$OUT/tr_$cur"
done
rocprofv3 --kernel-trace --output-format csv -d "$OUT/tr_r18" -- python3 scripts/engine_product_driver.py --workload resnet18 --products 12 --out "$OUT/launches_r18.json" > "$OUT/tr_r18.log" 2>&1
python3 scripts/product_trace_table.py "$OUT/launches_r18.json" "$OUT/tr_r18" > "$OUT/r03_resnet18_one_product_trace.txt"
rm -rf "$OUT/tr_r18"
ls -la "$OUT"

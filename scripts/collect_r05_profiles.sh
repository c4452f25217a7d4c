#!/bin/bash
# Round-5 evidence run (on the GPU box, from the repo root):  bash scripts/collect_r05_profiles.sh gpurun_out/r5p
# Everything the DESIGN.md section 6 tables of round 5 quote; raw profiler output is deleted (gpurun_out/ must stay small).
set -u
OUT=${1:-gpurun_out/r5p}
mkdir -p "$OUT"
export TMPDIR=/tmp
# the whole GPU suite with its tolerance log and durations
( time HF_TOL_LOG="$OUT/r05_tolerance_sites.jsonl" python -m pytest tests -q -m gpu --durations=15 ) > "$OUT/full_suite.log" 2>&1; echo "full_suite rc=$?" >> "$OUT/rc.log"
# headline, as the driver runs it
python bench.py --gpus 1 --steps 20 --warmup 5 > "$OUT/r05_bench_n1.json" 2> "$OUT/r05_bench_n1.err"; echo "bench_n1 rc=$?" >> "$OUT/rc.log"
# rocprofv3 summary of the same command (no CPU leg, no step / train-mode legs: kernels of the timed loop)
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_bench -- python3 bench.py --no-cpu-baseline --no-step-timing --no-beyond-l3 --no-train-bn > "$OUT/r05_bench_n1_under_rocprof.json" 2> "$OUT/prof_bench.log"
find /tmp/prof_bench -name "*kernel_stats.csv" -exec cp {} "$OUT/r05_bench_n1_kernel_stats.csv" \;
rm -rf /tmp/prof_bench
# PMC traffic of the PCG kernels (separate passes; FETCH_SIZE doubled per the guide): profiles/traffic.json
for C in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && timeout -s KILL 200 rocprofv3 --pmc $C --kernel-trace --output-format csv -d /tmp/pmck_$C -- python3 $GRAFT_REPO_ROOT/scripts/pcg_kernel_bench.py --sizes 11175370 --iters 12 > $GRAFT_REPO_ROOT/$OUT/pmck_$C.out 2> $GRAFT_REPO_ROOT/$OUT/pmck_$C.err; echo "pmc $C rc=$?" >> $GRAFT_REPO_ROOT/$OUT/rc.log)
done
python scripts/pmc_traffic.py /tmp/pmck_FETCH_SIZE /tmp/pmck_WRITE_SIZE 11175370 > "$OUT/r05_traffic.json" 2>> "$OUT/other.err"
rm -rf /tmp/pmck_FETCH_SIZE /tmp/pmck_WRITE_SIZE
# other workloads
: > "$OUT/r05_other_workloads.jsonl"
for args in "--workload allcnnc" "--workload allcnnc --curvature hessian --precond 1 --damping 1.0" "--workload resnet50" "--workload resnet18 --bn train" "--workload resnet18 --curvature hessian" "--workload resnet18 --bn train --curvature hessian" "--workload resnet18 --acc 16,16"; do
  python bench.py $args --steps 3 --warmup 1 >> "$OUT/r05_other_workloads.jsonl" 2>> "$OUT/other.err"
done
# data-parallel paths that one GPU can exercise
python bench.py --force-dist 1 --chunk 0 --no-cpu-baseline --no-step-timing > "$OUT/r05_bench_1rank_rccl.json" 2>> "$OUT/dp.err"
python bench.py --force-dist 1 --chunk 1 --no-cpu-baseline --no-step-timing > "$OUT/r05_bench_1rank_rccl_chunked.json" 2>> "$OUT/dp.err"
python bench.py --gpus 2 --steps 2 --no-cpu-baseline > "$OUT/r05_bench_2ranks_one_gpu_gloo.json" 2>> "$OUT/dp.err"
# PCG vector kernels at the vector sizes of the workloads
python scripts/pcg_kernel_bench.py > "$OUT/r05_pcg_kernel_bench.jsonl" 2>> "$OUT/other.err"
# convolution kernels on the large-map shapes (All-CNN-C, ResNet-50 topology)
python scripts/conv_kernel_bench.py --big 1 --no-reduce 1 > "$OUT/r05_conv_kernel_bench_big.jsonl" 2>> "$OUT/other.err"
ls -la "$OUT"; cat "$OUT/rc.log"

#!/bin/bash
# Round-4 evidence run (on the GPU box, from the repo root):  bash scripts/collect_r04_profiles.sh gpurun_out/r4p
set -u
OUT=${1:-gpurun_out/r4p}
mkdir -p "$OUT"
export TMPDIR=/tmp
# the whole GPU suite, with durations
( time python -m pytest tests -q -m gpu --durations=15 ) > "$OUT/full_suite.log" 2>&1; echo "full_suite rc=$?" >> "$OUT/rc.log"
# headline, as the driver runs it
python bench.py --gpus 1 --steps 20 --warmup 5 > "$OUT/r04_bench_n1.json" 2> "$OUT/r04_bench_n1.err"; echo "bench_n1 rc=$?" >> "$OUT/rc.log"
# rocprofv3 summary of the same command (no CPU leg, no step leg: kernels of the timed loop)
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_bench" -- python3 bench.py --no-cpu-baseline --no-step-timing --no-beyond-l3 > "$OUT/r04_bench_n1_under_rocprof.json" 2> "$OUT/prof_bench.log"
find "$OUT/prof_bench" -name "*kernel_stats.csv" -exec cp {} "$OUT/r04_bench_n1_kernel_stats.csv" \;
rm -rf "$OUT/prof_bench"
# PMC traffic of the PCG kernels (separate passes; FETCH_SIZE doubled per the guide): profiles/traffic.json
for C in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && timeout -s KILL 200 rocprofv3 --pmc $C --kernel-trace --output-format csv -d /tmp/pmck_$C -- python3 $GRAFT_REPO_ROOT/scripts/pcg_kernel_bench.py --sizes 11175370 --iters 12 > $GRAFT_REPO_ROOT/$OUT/pmck_$C.out 2> $GRAFT_REPO_ROOT/$OUT/pmck_$C.err; echo "pmc $C rc=$?" >> $GRAFT_REPO_ROOT/$OUT/rc.log)
done
python scripts/pmc_traffic.py /tmp/pmck_FETCH_SIZE /tmp/pmck_WRITE_SIZE 11175370 > "$OUT/r04_traffic.json" 2>> "$OUT/other.err"
# other workloads
: > "$OUT/r04_other_workloads.jsonl"
for args in "--workload allcnnc" "--workload allcnnc --curvature hessian --precond 1 --damping 1.0" "--workload resnet50" "--workload resnet18 --bn train" "--workload resnet18 --curvature hessian" "--workload resnet18 --acc 16,16"; do
  python bench.py $args --steps 3 --warmup 1 >> "$OUT/r04_other_workloads.jsonl" 2>> "$OUT/other.err"
done
: > "$OUT/r04_resnet50_conv_blocks.jsonl"
for blocks in 384 512 768; do
  echo "== HF_CONV_BLOCKS=$blocks" >> "$OUT/r04_resnet50_conv_blocks.jsonl"
  HF_CONV_BLOCKS=$blocks python bench.py --workload resnet50 --steps 2 --warmup 1 --no-cpu-baseline --no-step-timing >> "$OUT/r04_resnet50_conv_blocks.jsonl" 2>> "$OUT/other.err"
done
: > "$OUT/r04_autograd_paths.jsonl"
for args in "--workload resnet18 --curvature hessian" "--workload resnet18 --bn train"; do
  HF_ENGINE=0 python bench.py $args --steps 2 --warmup 1 --no-cpu-baseline --no-step-timing >> "$OUT/r04_autograd_paths.jsonl" 2>> "$OUT/other.err"
done
# data-parallel paths that one GPU can exercise
python bench.py --force-dist 1 --chunk 0 --no-cpu-baseline --no-step-timing > "$OUT/r04_bench_1rank_rccl.json" 2>> "$OUT/dp.err"
python bench.py --force-dist 1 --chunk 1 --no-cpu-baseline --no-step-timing > "$OUT/r04_bench_1rank_rccl_chunked.json" 2>> "$OUT/dp.err"
python bench.py --force-dist 1 --no-cpu-baseline --no-step-timing > "$OUT/r04_bench_1rank_rccl_auto.json" 2>> "$OUT/dp.err"
python bench.py --gpus 2 --steps 2 --no-cpu-baseline > "$OUT/r04_bench_2ranks_one_gpu_gloo.json" 2>> "$OUT/dp.err"
python bench.py --gpus 8 --batch 4 --steps 1 --warmup 1 --iters 60 --no-cpu-baseline > "$OUT/r04_bench_8ranks_one_gpu_gloo.json" 2>> "$OUT/dp.err"
# PCG vector kernels at the vector sizes of the workloads
python scripts/pcg_kernel_bench.py > "$OUT/r04_pcg_kernel_bench.jsonl" 2>> "$OUT/other.err"
# engine kernel counters (k_pack's LDS conflicts after the swizzle, the convolutions) -- ResNet-18 product
bash scripts/run_engine_counters.sh "$OUT/pmc" > "$OUT/pmc.log" 2>&1
cp "$OUT/pmc/engine_kernel_counters.json" "$OUT/r04_engine_kernel_counters.json" 2>/dev/null
# one-product traces: ResNet-18 train mode, Hessian, ResNet-50
for spec in "resnet18 ggn train r18_train" "resnet18 hessian eval r18_hessian" "resnet50 ggn eval r50"; do
  set -- $spec
  rocprofv3 --kernel-trace --output-format csv -d "$OUT/tr_$4" -- python3 scripts/engine_product_driver.py --workload $1 --curvature $2 --bn $3 --products 8 --out "$OUT/launches_$4.json" > "$OUT/tr_$4.log" 2>&1
  python3 scripts/product_trace_table.py "$OUT/launches_$4.json" "$OUT/tr_$4" > "$OUT/r04_$4_one_product_trace.txt" 2>> "$OUT/tr_$4.log"
  rm -rf "$OUT/tr_$4"
done
ls -la "$OUT"; cat "$OUT/rc.log"

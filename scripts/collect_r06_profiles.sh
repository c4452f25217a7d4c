#!/bin/bash
# Round-6 evidence run (on the GPU box, from the repo root):  bash scripts/collect_r06_profiles.sh gpurun_out/r6p
# Everything DESIGN.md section 6 quotes for round 6; raw profiler output is deleted (gpurun_out/ must stay small).
set -u
OUT=${1:-gpurun_out/r6p}
mkdir -p "$OUT"
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
# headline, as the driver runs it
python bench.py --gpus 1 --steps 20 --warmup 5 > "$OUT/r06_bench_n1.json" 2> "$OUT/r06_bench_n1.err"; echo "bench_n1 rc=$?" >> "$OUT/rc.log"
# rocprofv3 summary of the same command (no CPU leg, no step / train-mode legs: kernels of the timed loop)
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_bench -- python3 bench.py --no-cpu-baseline --no-step-timing --no-beyond-l3 --no-train-bn > "$OUT/r06_bench_n1_under_rocprof.json" 2> "$OUT/prof_bench.log"
find /tmp/prof_bench -name "*kernel_stats.csv" -exec cp {} "$OUT/r06_bench_n1_kernel_stats.csv" \;
rm -rf /tmp/prof_bench
# PMC traffic of the PCG kernels (separate passes; FETCH_SIZE doubled per the guide): the workload's N without and WITH
# the diagonal preconditioner, and config 4's N with it -> profiles/traffic.json keys k_*_<N>[_precond]
: > "$OUT/traffic_parts.jsonl"
for spec in "11175370 0" "11175370 1" "1387108 1"; do
  set -- $spec; N=$1; P=$2; SUF=""; [ "$P" = "1" ] && SUF="_precond"
  for C in FETCH_SIZE WRITE_SIZE; do
    (cd /tmp && timeout -s KILL 200 rocprofv3 --pmc $C --kernel-trace --output-format csv -d /tmp/pmck_$C -- python3 $R/scripts/pcg_kernel_bench.py --sizes $N --iters 12 --precond $P > $R/$OUT/pmck_${C}_${N}_$P.out 2> $R/$OUT/pmck_${C}_${N}_$P.err; echo "pmc $C $N $P rc=$?" >> $R/$OUT/rc.log)
  done
  python scripts/pmc_traffic.py /tmp/pmck_FETCH_SIZE /tmp/pmck_WRITE_SIZE $N $SUF | python -c "import json,sys; print(json.dumps(json.load(sys.stdin)))" >> "$OUT/traffic_parts.jsonl" 2>> "$OUT/other.err"
  rm -rf /tmp/pmck_FETCH_SIZE /tmp/pmck_WRITE_SIZE
done
python - "$OUT" <<'PY'
import json, sys
out = sys.argv[1]
merged = {"raw_KB": {}, "launches": {}}
for ln in open(f"{out}/traffic_parts.jsonl"):
    d = json.loads(ln)
    keys = [k for k in d if k not in ("raw_KB", "launches")]
    tag = next((k[len("k_update_xr_"):] for k in keys if k.startswith("k_update_xr_")), "?")
    for k in keys:
        merged[k] = d[k]
    merged["raw_KB"][tag] = d["raw_KB"]
    merged["launches"][tag] = d["launches"]
json.dump(merged, open(f"{out}/r06_traffic.json", "w"), indent=1)
PY
# other workloads (each with step_ms)
: > "$OUT/r06_other_workloads.jsonl"
for args in "--workload allcnnc" "--workload allcnnc --curvature hessian --precond 1 --damping 1.0" "--workload resnet50" "--workload resnet18 --bn train" "--workload resnet18 --curvature hessian" "--workload resnet18 --bn train --curvature hessian" "--workload resnet18 --acc 16,16" "--workload resnet18 --freeze stem+layer1"; do
  python bench.py $args --steps 3 --warmup 1 --no-train-bn >> "$OUT/r06_other_workloads.jsonl" 2>> "$OUT/other.err"
done
# data-parallel paths that one GPU can exercise: 1-rank RCCL group (both forms), 2 and 8 rank processes over gloo through
# the fallback ladder (own launcher and under torch.distributed.run, as the driver starts it), fault injection
python bench.py --force-dist 1 --chunk 0 --no-cpu-baseline --no-step-timing > "$OUT/r06_bench_1rank_rccl.json" 2>> "$OUT/dp.err"
python bench.py --force-dist 1 --chunk 1 --no-cpu-baseline --no-step-timing > "$OUT/r06_bench_1rank_rccl_chunked.json" 2>> "$OUT/dp.err"
python bench.py --gpus 2 --steps 2 --no-cpu-baseline > "$OUT/r06_bench_2ranks_one_gpu_gloo.json" 2>> "$OUT/dp.err"
python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29631 bench.py --gpus 2 --steps 2 --warmup 1 --no-cpu-baseline > "$OUT/r06_bench_2ranks_torchrun_one_gpu_gloo.json" 2>> "$OUT/dp.err"
python bench.py --gpus 8 --steps 2 --warmup 1 --no-cpu-baseline > "$OUT/r06_bench_8ranks_one_gpu_gloo.json" 2>> "$OUT/dp.err"
HF_TEST_DP_FAULT=raise:not_plain python bench.py --gpus 2 --steps 2 --warmup 1 --no-cpu-baseline > "$OUT/r06_bench_2ranks_ladder_fault_injected.json" 2>> "$OUT/dp.err"
# PCG vector kernels at the vector sizes of the workloads, without / with the diagonal preconditioner
python scripts/pcg_kernel_bench.py --sizes 1387108,11175370,25557032,67108864,100000000 > "$OUT/r06_pcg_kernel_bench.jsonl" 2>> "$OUT/other.err"
python scripts/pcg_kernel_bench.py --sizes 1387108,11175370,25557032 --precond 1 >> "$OUT/r06_pcg_kernel_bench.jsonl" 2>> "$OUT/other.err"
# the whole GPU suite with its tolerance log and durations
( time HF_TOL_LOG="$OUT/r06_tolerance_sites.jsonl" python -m pytest tests -q -m gpu --durations=15 -p no:cacheprovider ) > "$OUT/full_suite.log" 2>&1; echo "full_suite rc=$?" >> "$OUT/rc.log"
rm -f "$OUT"/pmck_*.out
ls -la "$OUT"; cat "$OUT/rc.log"

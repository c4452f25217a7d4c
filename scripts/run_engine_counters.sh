#!/bin/bash
# Four rocprofv3 passes over scripts/engine_product_driver.py (the program itself after `--`): kernel trace,
# FETCH_SIZE, WRITE_SIZE (they do not fit one pass) and an SQ pass; then the joined table.
# Usage (on the GPU box, from the repo root):  bash scripts/run_engine_counters.sh gpurun_out/r3/pmc
set -u
OUT=${1:-gpurun_out/pmc}
mkdir -p "$OUT"
export TMPDIR=/tmp
# DRV_ARGS: extra driver arguments (e.g. "--workload resnet50", "--bn train")
DRV="python3 scripts/engine_product_driver.py --products ${DRV_PRODUCTS:-20} ${DRV_ARGS:-}"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- $DRV --out "$OUT/launches.json" > "$OUT/trace.log" 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch" -- $DRV > "$OUT/fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/write" -- $DRV > "$OUT/write.log" 2>&1
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY \
  --output-format csv -d "$OUT/sq" -- $DRV > "$OUT/sq.log" 2>&1
python3 scripts/pmc_engine_table.py "$OUT/launches.json" "$OUT/trace" "$OUT/fetch" "$OUT/write" "$OUT/sq" > "$OUT/engine_kernel_counters.json" 2> "$OUT/table.err"
tail -3 "$OUT"/*.log "$OUT/table.err"
# keep only the small artefacts (the raw CSVs are tens of MB)
find "$OUT" -name "*.csv" -size +2M -delete

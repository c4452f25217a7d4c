#!/bin/bash
# Round-6 A/B on one box: how many partial rows per thread the consumers of a train-mode BatchNorm keep in flight per
# round trip (HF_FCS_BATCH in csrc/hf_bn.hip::final_column_sums2; 4 = rounds 4-5), whole bench with --bn train;
# variants under build_variants/.
OUT=${1:-gpurun_out/fcs}; mkdir -p $OUT
: > $OUT/fcs_batch_ab.jsonl
for rep in 1 2; do
  for lib in "" $PWD/build_variants/libhfpcg_fcs8.so $PWD/build_variants/libhfpcg_fcs16.so; do
    for args in "--bn train" "--bn train --curvature hessian"; do
      echo "== HF_PCG_LIB=$lib $args" >> $OUT/fcs_batch_ab.jsonl
      HF_PCG_LIB=$lib python bench.py $args --steps 4 --warmup 2 --no-cpu-baseline --no-step-timing --no-beyond-l3 --no-train-bn >> $OUT/fcs_batch_ab.jsonl 2>> $OUT/err.log
    done
  done
done
python - $OUT/fcs_batch_ab.jsonl <<'PY'
import json, sys
lib = None
for ln in open(sys.argv[1]):
    if ln.startswith("=="):
        lib = ln.strip()
    elif ln.startswith("{"):
        d = json.loads(ln)
        print(lib, round(d["value"], 1))
PY

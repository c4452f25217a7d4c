#!/bin/bash
# Round-6 A/B on one box: column-blocked workgroup mapping of the train-mode BatchNorm's elementwise passes
# (HF_AFF_COLB in csrc/hf_bn.hip; off = the row-major mapping of rounds 4-5), whole bench with --bn train, and the
# train-mode tests (the summation order of the per-channel sums changes: tolerances, not bits).
OUT=${1:-gpurun_out/colb}; mkdir -p $OUT
python -m pytest tests/test_engine_gpu.py tests/test_optimizer_gpu.py tests/test_session_gpu.py tests/test_acc_session_gpu.py -m gpu -q -p no:cacheprovider -k "train or frozen or mse" > $OUT/train_tests.log 2>&1
tail -3 $OUT/train_tests.log
: > $OUT/colb_ab.jsonl
for rep in 1 2; do
  for lib in "" $PWD/build_variants/libhfpcg_colb_off.so; do
    for args in "--bn train" "--bn train --curvature hessian"; do
      echo "== HF_PCG_LIB=$lib $args" >> $OUT/colb_ab.jsonl
      HF_PCG_LIB=$lib python bench.py $args --steps 4 --warmup 2 --no-cpu-baseline --no-step-timing --no-beyond-l3 --no-train-bn >> $OUT/colb_ab.jsonl 2>> $OUT/err.log
    done
  done
done
python - $OUT/colb_ab.jsonl <<'PY'
import json, sys
lib = None
for ln in open(sys.argv[1]):
    if ln.startswith("=="):
        lib = ln.strip()
    elif ln.startswith("{"):
        d = json.loads(ln)
        print(lib, round(d["value"], 1))
PY
tail -c 300 $OUT/err.log

#!/bin/bash
# (experiment) bytes of the activation map per workgroup of the BatchNorm adjoint's row-major kernel
# (engine/buffers.py::_Buffers.ADJ_BYTES_PER_WG, 32768 since round 3), whole bench on the large-map workloads.
OUT=${1:-gpurun_out/adjgrid}; mkdir -p $OUT
: > $OUT/adj_grid_ab.jsonl
for rep in 1 2; do
  for bytes in 32768 16384 8192; do
    for args in "--workload resnet50" "--workload allcnnc" "--workload resnet18"; do
      echo "== ADJ_BYTES_PER_WG=$bytes $args" >> $OUT/adj_grid_ab.jsonl
      python -c "
import sys, runpy
import pytorchhessianfree_amd.engine.buffers as b
b._Buffers.ADJ_BYTES_PER_WG = $bytes
sys.argv = ['bench.py'] + '$args --steps 3 --warmup 1 --no-cpu-baseline --no-step-timing --no-beyond-l3 --no-train-bn'.split()
runpy.run_path('bench.py', run_name='__main__')
" >> $OUT/adj_grid_ab.jsonl 2>> $OUT/err.log
    done
  done
done
python - $OUT/adj_grid_ab.jsonl <<'PY'
import json, sys
tag = None
for ln in open(sys.argv[1]):
    if ln.startswith("=="):
        tag = ln.strip()
    elif ln.startswith("{"):
        print(tag, round(json.loads(ln)["value"], 1))
PY
tail -c 300 $OUT/err.log

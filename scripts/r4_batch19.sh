#!/bin/bash
# Round-4 GPU batch 19: train-mode pair launches for the downsample blocks -- parity, then A/B.
O=gpurun_out/r4u; mkdir -p $O
timeout 900 python -m pytest tests/test_engine_gpu.py tests/test_session_gpu.py tests/test_acc_session_gpu.py -q -m gpu -k "train_mode or folded" -x > $O/tests.log 2>&1; echo "tests rc=$?" >> $O/rc.log
tail -5 $O/tests.log
if grep -q "tests rc=0" $O/rc.log; then
  for rep in 1 2; do
    for pair in 1 0; do
      echo "== HF_BN_TRAIN_PAIR=$pair" >> $O/train_pair.jsonl
      HF_BN_TRAIN_PAIR=$pair timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --bn train --no-beyond-l3 >> $O/train_pair.jsonl 2>> $O/train.err
    done
  done
  python - <<'PY'
import json
for l in open("gpurun_out/r4u/train_pair.jsonl"):
    if l.startswith("{"):
        d=json.loads(l); print("   ", round(d["value"],1), round(d["step_ms"]["mean"],2))
    else: print(l.strip())
PY
fi

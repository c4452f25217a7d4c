#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_optimizer_gpu.py -q -x -k "pack or unpack or scatter" 2>&1 | tail -2
timeout 300 python tests/gpu_workers/nhwc_small_net.py 2>&1 | grep -E "RESULT|Error|error" | cut -c1-200
for i in 1 2; do
timeout 900 python bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | grep -a '^{"metric"' | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print(round(d['value'],1), d['config']['matvec'][-100:])"
done
rm -rf gpurun_out/prof_r17
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r17 -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/prof_r17.log 2>&1
python3 - <<P
import csv,glob,os
f=sorted(glob.glob("gpurun_out/prof_r17/*/*kernel_stats.csv"),key=os.path.getmtime)[-1]
for r in csv.DictReader(open(f)):
    if "k_pack<float" in r["Name"] or "k_unpack" in r["Name"]: print(r["Calls"].rjust(6), "%7.2f"%(float(r["AverageNs"])/1e3), r["Name"][:60])
P
find gpurun_out/prof_r17 -name "*kernel_trace.csv" -delete

#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/ -m gpu -q 2>&1 | tail -3
timeout 600 python __graft_entry__.py smoke 2>&1 | tail -1
timeout 900 python bench.py > gpurun_out/bench_r1_final5.json 2> gpurun_out/bench_r1_final5.err; grep -a "\[bench\]" gpurun_out/bench_r1_final5.err; cut -c1-400 gpurun_out/bench_r1_final5.json
rm -rf gpurun_out/prof_final2
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_final2 -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/prof_final2.log 2>&1
grep -a "\[bench\]" gpurun_out/prof_final2.log; grep -a -o '"value": [0-9.]*' gpurun_out/prof_final2.log; grep -a -o 'N[HC][WH][CW] [^"]*"' gpurun_out/prof_final2.log
find gpurun_out/prof_final2 -name "*kernel_trace.csv" -delete
rm -rf gpurun_out/miopen_db_after; cp -r pytorchhessianfree_amd/miopen_db gpurun_out/miopen_db_after

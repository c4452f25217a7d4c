#!/bin/bash
# round 2, GPU session 1: sanity of the new fused-iteration graph, full GPU tests, bench,
# MIOpen no-split-K experiment, small-N kernel bench
set -x
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r2s1; mkdir -p $O
export TMPDIR=/tmp
timeout 300 python __graft_entry__.py smoke > $O/smoke.log 2>&1; echo "smoke rc=$?"
timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_fused.json 2> $O/bench_fused.err; echo "bench rc=$?"
HF_FUSE_ITERATION=0 timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_unfused.json 2> $O/bench_unfused.err; echo "bench rc=$?"
tail -c 600 $O/bench_fused.err
python scripts/miopen_nogks_db.py pytorchhessianfree_amd/miopen_db /tmp/nogks_db
MIOPEN_USER_DB_PATH=/tmp/nogks_db timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_nogks.json 2> $O/bench_nogks.err; echo "bench nogks rc=$?"
tail -c 600 $O/bench_nogks.err
(cd /tmp && MIOPEN_USER_DB_PATH=/tmp/nogks_db timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_nogks -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/$O/prof_nogks.json 2> $GRAFT_REPO_ROOT/$O/prof_nogks.err)
f=$(find /tmp/prof_nogks -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -40 "$f" | cut -c1-200 > $O/prof_nogks_kernel_stats_head.csv
timeout 300 python scripts/pcg_kernel_bench.py > $O/kbench.jsonl 2>&1
timeout 300 python scripts/pcg_kernel_bench.py --precond 1 >> $O/kbench.jsonl 2>&1
timeout 1500 python -m pytest tests -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc=$?"
tail -30 $O/pytest.log
cat $O/bench_fused.json | cut -c1-1500
cat $O/bench_unfused.json | cut -c1-300
cat $O/bench_nogks.json | cut -c1-1500
cat $O/kbench.jsonl

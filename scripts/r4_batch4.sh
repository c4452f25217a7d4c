#!/bin/bash
# Round-4 GPU batch 4: clean rocprofv3 summary of the bench, NT A/B in one box, new tests, chained two-phase product.
O=gpurun_out/r4e; mkdir -p $O
export TMPDIR=/tmp
run() { name=$1; shift; "$@" > $O/$name.log 2>&1; echo "$name rc=$?" >> $O/rc.log; }
run newtests python -m pytest tests/test_distributed_gpu.py tests/test_session_gpu.py -q -m gpu -k "acc_step_two_ranks or one_launch or bottleneck or measured"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bench -- python3 bench.py --no-cpu-baseline --no-step-timing --no-beyond-l3 > $O/r04_bench_n1_under_rocprof.json 2> $O/prof_bench.log
find $O/prof_bench -name "*kernel_stats.csv" -exec cp {} $O/r04_bench_n1_kernel_stats.csv \;
rm -rf $O/prof_bench
: > $O/nt_ab.jsonl
for rep in 1 2; do
  for nt in 4000000 999999999999; do
    echo "== HF_PCG_NT_MIN=$nt rep $rep" >> $O/nt_ab.jsonl
    HF_PCG_NT_MIN=$nt python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-step-timing --no-beyond-l3 >> $O/nt_ab.jsonl 2>> $O/nt_ab.err
  done
done
python bench.py --force-dist 1 --chunk 1 --no-cpu-baseline --no-step-timing --no-beyond-l3 > $O/r04_bench_1rank_rccl_chunked.json 2>> $O/dp.err
HF_CHUNK_ONEGRAPH=1 python bench.py --force-dist 1 --chunk 1 --no-cpu-baseline --no-step-timing --no-beyond-l3 > $O/r04_bench_1rank_rccl_chunked_one_launch.json 2>> $O/dp.err
python bench.py --force-dist 1 --no-cpu-baseline --no-step-timing --no-beyond-l3 > $O/r04_bench_1rank_rccl_auto.json 2>> $O/dp.err
rocprofv3 --kernel-trace --output-format csv -d $O/trace_chain -- python3 bench.py --force-dist 1 --chunk 1 --steps 1 --warmup 1 --iters 40 --no-cpu-baseline --no-step-timing --no-beyond-l3 > $O/trace_chain.json 2> $O/trace_chain.err
HF_CHUNK_ONEGRAPH=1 rocprofv3 --kernel-trace --output-format csv -d $O/trace_chain1 -- python3 bench.py --force-dist 1 --chunk 1 --steps 1 --warmup 1 --iters 40 --no-cpu-baseline --no-step-timing --no-beyond-l3 > $O/trace_chain1.json 2> $O/trace_chain1.err
python scripts/iteration_trace_table.py $O/trace_chain > $O/r04_chunked_iteration_trace.txt 2>&1
python scripts/iteration_trace_table.py $O/trace_chain1 > $O/r04_chunked_one_launch_iteration_trace.txt 2>&1
rm -rf $O/trace_chain $O/trace_chain1
cat $O/rc.log

#!/bin/bash
# Round-4 GPU batch 1: new parity tests, K1 variants, config-4 / Hessian bench lines, 1-rank RCCL traces.
O=gpurun_out/r4c; mkdir -p $O
run() { name=$1; shift; "$@" > $O/$name.log 2>&1; echo "$name rc=$?" >> $O/rc.log; }
run acc      python -m pytest tests/test_acc_session_gpu.py -q -m gpu
run dist     python -m pytest tests/test_distributed_gpu.py -q -m gpu -k "eight or refused or rccl or launcher or measured or bench_gpus_2"
run session  python -m pytest tests/test_session_gpu.py -q -m gpu -k "allcnnc or bottleneck or reverified or train_mode"
run hessian  python -m pytest tests/test_engine_gpu.py -q -m gpu -k "hessian or train_mode"
run config4  python -m pytest tests/test_optimizer_gpu.py -q -m gpu -k "config4"
for v in main u1_4 k1nt k1nt_u1_4; do
  lib=pytorchhessianfree_amd/csrc/variants/libhfpcg_$v.so
  [ $v = main ] && lib=pytorchhessianfree_amd/csrc/libhfpcg.so
  echo "== $v" >> $O/k1_variants.jsonl
  HF_PCG_LIB=$PWD/$lib python scripts/pcg_kernel_bench.py --sizes 11175370,100000000 >> $O/k1_variants.jsonl 2>> $O/k1_variants.err
done
echo "== main HF_PCG_BLOCKS=1024" >> $O/k1_variants.jsonl
HF_PCG_BLOCKS=1024 python scripts/pcg_kernel_bench.py --sizes 11175370,100000000 >> $O/k1_variants.jsonl 2>> $O/k1_variants.err
python bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_n1.json 2> $O/bench_n1.err; echo "bench_n1 rc=$?" >> $O/rc.log
python bench.py --steps 3 --warmup 1 --no-cpu-baseline --workload allcnnc --curvature hessian --precond 1 > $O/bench_config4.json 2> $O/bench_config4.err; echo "bench_config4 rc=$?" >> $O/rc.log
python bench.py --steps 3 --warmup 1 --no-cpu-baseline --bn train > $O/bench_train.json 2> $O/bench_train.err; echo "bench_train rc=$?" >> $O/rc.log
python bench.py --steps 3 --warmup 1 --no-cpu-baseline --workload resnet18 --curvature hessian > $O/bench_r18_hessian.json 2> $O/bench_r18_hessian.err; echo "bench_r18_hessian rc=$?" >> $O/rc.log
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/trace_chunk -- python3 $GRAFT_REPO_ROOT/bench.py --force-dist 1 --chunk 1 --steps 1 --warmup 1 --iters 40 --no-cpu-baseline --no-step-timing > $GRAFT_REPO_ROOT/$O/trace_chunk.json 2> $GRAFT_REPO_ROOT/$O/trace_chunk.err
echo "trace_chunk rc=$?" >> $GRAFT_REPO_ROOT/$O/rc.log
cd $GRAFT_REPO_ROOT
# keep only the kernel trace csv (small) -- drop anything large
find $O/trace_chunk -type f ! -name "*kernel_trace.csv" -delete 2>/dev/null
ls -la $O/trace_chunk/*/* 2>/dev/null | head
cat $O/rc.log

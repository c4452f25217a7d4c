cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r2pmc; mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for C in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && timeout -s KILL 150 rocprofv3 --pmc $C --kernel-trace --output-format csv -d /tmp/pmck_$C -- python3 $R/scripts/pcg_kernel_bench.py --sizes 11175370 --iters 12 > $R/$O/pmck_$C.out 2> $R/$O/pmck_$C.err; echo "pmc kernel bench $C rc=$?")
  tail -c 300 $O/pmck_$C.err | tr '\n' ' '; echo
done
python scripts/pmc_traffic.py /tmp/pmck_FETCH_SIZE /tmp/pmck_WRITE_SIZE 11175370 > $O/traffic_kbench.json; head -c 900 $O/traffic_kbench.json

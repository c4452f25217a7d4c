#!/bin/bash
# Same-box A/B of the LDS swizzle key of k_pack's permuting path: the headline bench and the rocprofv3 average of
# k_pack, two repetitions (profiles/r05_pack_swizzle_ab.jsonl).  Variant 1 -- conflict-free stores: the key of the
# in-quad XOR is  __umulhi(word, 0xFFFFFFFFu / (32 * HW) + 1)  instead of  word >> 6 , in the staging store and in the
# 16-byte read that undoes it -- measured neutral and is NOT in the tree; the script expects the two libraries as
# build_variants/libhfpcg_swz{0,1}.so.
OUT=${1:-gpurun_out/r5swz}; mkdir -p $OUT; export TMPDIR=/tmp
: > $OUT/ab.jsonl
for rep in 1 2; do for v in 0 1; do
  HF_PCG_LIB=$PWD/build_variants/libhfpcg_swz$v.so python bench.py --no-cpu-baseline --no-step-timing --no-train-bn --no-beyond-l3 > $OUT/b.json 2>$OUT/b.err
  python - $v $rep $OUT <<'PY' >> $OUT/ab.jsonl
import json,sys
d=json.loads(open(sys.argv[3]+"/b.json").read().strip().splitlines()[-1]); print(json.dumps({"HF_PACK_SWZ_RUN":int(sys.argv[1]),"rep":int(sys.argv[2]),"matvecs_per_s":round(d["value"],1)}))
PY
done; done
for v in 0 1; do
  HF_PCG_LIB=$PWD/build_variants/libhfpcg_swz$v.so rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pswz$v -- python3 bench.py --no-cpu-baseline --no-step-timing --no-train-bn --no-beyond-l3 --steps 1 --warmup 0 > /dev/null 2>$OUT/p$v.err
  f=$(find /tmp/pswz$v -name "*kernel_stats.csv" | head -1)
  python - $v "$f" <<'PY' >> $OUT/ab.jsonl
import csv,json,sys
for r in csv.DictReader(open(sys.argv[2])):
    if "k_pack" in r["Name"]: print(json.dumps({"HF_PACK_SWZ_RUN":int(sys.argv[1]),"kernel":"k_pack","calls":int(r["Calls"]),"avg_us":round(float(r["AverageNs"])/1e3,2)}))
PY
  rm -rf /tmp/pswz$v
done
cat $OUT/ab.jsonl

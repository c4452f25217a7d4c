# ordered kernel list of ONE product (the last graph replay of scripts/one_product_trace.py)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r2final4; mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
D=/tmp/prof_t_$$; rm -rf $D
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d $D -- python3 $R/scripts/one_product_trace.py > /dev/null 2> $R/$O/trace.err)
f=$(find $D -name "*kernel_trace.csv" | head -1)
python - "$f" "$O/one_product_trace.txt" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
last = max(i for i, r in enumerate(rows) if "k_unpack_tangent" in r["Kernel_Name"])
prod = rows[last:]
stop = min(i for i, r in enumerate(prod) if "k_pack<" in r["Kernel_Name"])
prod = prod[:stop + 1]  # unpack ... pack = one product
t0 = int(prod[0]["Start_Timestamp"])
with open(sys.argv[2], "w") as f:
    f.write("# kernels of ONE GGN product (ResNet-18 workload, fused engine, hipGraph replay), in start order\n")
    f.write("# start_us  duration_us  kernel\n")
    for r in prod:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        f.write(f"{(s - t0) / 1e3:9.2f} {(e - s) / 1e3:8.2f}  {r['Kernel_Name'][:110]}\n")
    f.write(f"# {len(prod)} launches, {(int(prod[-1]['End_Timestamp']) - t0) / 1e3:.1f} us from first start to last end\n")
print(open(sys.argv[2]).read()[-1500:])
PY

"""Per-launch table of one engine product: kernel-trace CSV (rocprofv3 --kernel-trace on
scripts/engine_product_driver.py) joined by dispatch order with the driver's algorithmic bytes / flops.
    python scripts/product_trace_table.py <launches.json> <trace dir>"""
import csv, glob, json, os, sys
L = json.load(open(sys.argv[1]))
rows = []
for p in glob.glob(os.path.join(sys.argv[2], "**", "*kernel_trace.csv"), recursive=True):
    rows += list(csv.DictReader(open(p)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
n = L["launches_per_product"]
# the library's own kernels (anonymous namespace, k_*) are what the driver logged; launches of other libraries inside
# a product (rocBLAS / ATen kernels of a classifier head the engine does not fuse) are listed by name only
ours = [r for r in rows if "(anonymous namespace)::k_" in r["Kernel_Name"]]
tail = ours[-n:]
t0 = int(tail[0]["Start_Timestamp"])
others = [r for r in rows if int(r["Start_Timestamp"]) >= t0 and "(anonymous namespace)::k_" not in r["Kernel_Name"]]
tot = 0.0
for i, (r, l) in enumerate(zip(tail, L["launches"])):
    want = l["kernel"].split("<")[0].split("(")[0]
    if want not in r["Kernel_Name"]:
        raise SystemExit(f"launch {i}: the driver logged {l['kernel']}, the trace has {r['Kernel_Name'][:80]}")
    us = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    tot += us
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "")[:52]
    mb = (l["read"] + l["written"]) / 1e6
    print(f"{name:52s} {us:8.1f} us  alg {mb:8.1f} MB {mb / us * 1e3 if us else 0:8.0f} GB/s {l['flops'] / us / 1e6 if us else 0:6.1f} TF  "
          f"wgs {int(r.get('Grid_Size', 0)) // max(1, int(r.get('Workgroup_Size', 1)))}")
print("sum of kernel durations per product: %.1f us, %d launches" % (tot, n))
if others:
    o_us = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in others) / 1e3
    names = sorted({r["Kernel_Name"].split("(")[0][:48] for r in others})
    print("+ %d launches of other libraries inside this product, %.1f us: %s" % (len(others), o_us, "; ".join(names)))

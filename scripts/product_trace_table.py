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
tail = rows[-n:]
tot = 0.0
for r, l in zip(tail, L["launches"]):
    us = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    tot += us
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "")[:52]
    mb = (l["read"] + l["written"]) / 1e6
    print(f"{name:52s} {us:8.1f} us  alg {mb:8.1f} MB {mb / us * 1e3 if us else 0:8.0f} GB/s {l['flops'] / us / 1e6 if us else 0:6.1f} TF  "
          f"wgs {int(r.get('Grid_Size', 0)) // max(1, int(r.get('Workgroup_Size', 1)))}")
print("sum of kernel durations per product: %.1f us, %d launches" % (tot, n))

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
def base(kernel):
    return kernel.split("<")[0].split("(")[0]


# the LAST product: from the last launch of its first logged kernel on; logged launches matched in order by name
anchor = base(L["launches"][0]["kernel"])
start = max(i for i, r in enumerate(ours) if anchor in r["Kernel_Name"])
tail, unlogged, i = [], [], start
for l in L["launches"]:
    while i < len(ours) and base(l["kernel"]) not in ours[i]["Kernel_Name"]:
        unlogged.append(ours[i])
        i += 1
    if i >= len(ours):
        raise SystemExit(f"logged launch {l['kernel']} not found in the trace")
    tail.append(ours[i])
    i += 1
t0 = int(tail[0]["Start_Timestamp"])
others = [r for r in rows if int(r["Start_Timestamp"]) >= t0 and "(anonymous namespace)::k_" not in r["Kernel_Name"]]
others += unlogged
tot = 0.0
for r, l in zip(tail, L["launches"]):
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
    print("+ %d launches without a cost model inside this product (other libraries' kernels of an unfused classifier "
          "head, copies), %.1f us: %s" % (len(others), o_us, "; ".join(names)))

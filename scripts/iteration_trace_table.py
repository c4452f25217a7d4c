"""One PCG iteration out of a rocprofv3 --kernel-trace CSV of bench.py: every kernel between two consecutive
``k_update_p`` launches (start offset, gap to the previous kernel's end, duration, queue, name).

    python scripts/iteration_trace_table.py <dir with *kernel_trace.csv> [which iteration from the end, default 12]
"""
import csv
import glob
import os
import sys


def main():
    root = sys.argv[1]
    back = int(sys.argv[2]) if len(sys.argv) > 2 else 12
    path = (glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True) or [root])[0]
    rows = sorted(csv.DictReader(open(path)), key=lambda r: int(r["Start_Timestamp"]))
    idx = [i for i, r in enumerate(rows) if "k_update_p" in r["Kernel_Name"]]
    a, b = idx[-back], idx[-back + 1]
    t0 = prev = int(rows[a]["End_Timestamp"])
    span = (int(rows[b]["End_Timestamp"]) - t0) / 1e3
    print(f"# one PCG iteration (K3 end -> next K3 end): {span:.1f} us, {b - a} kernels; source: {os.path.basename(path)}")
    print(f"# {'start':>8} {'gap':>6} {'dur':>6}  queue  kernel")
    busy = 0.0
    for r in rows[a + 1:b + 1]:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
        print(f"{(s - t0) / 1e3:10.1f} {(s - prev) / 1e3:6.1f} {(e - s) / 1e3:6.1f}  q{r['Queue_Id']:<4} {name[:70]}")
        prev = max(prev, e)
        busy += (e - s) / 1e3
    print(f"# sum of kernel durations {busy:.1f} us, idle {span - busy:.1f} us")


if __name__ == "__main__":
    main()

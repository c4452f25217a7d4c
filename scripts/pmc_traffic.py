"""Turn two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs, as
MI355X_MICROARCH.md prescribes: the two counters do not fit one pass) into the per-launch HBM
traffic table ``profiles/traffic.json`` that bench.py quotes in ``roofline.traffic``.

    python scripts/pmc_traffic.py <dir of FETCH_SIZE pass> <dir of WRITE_SIZE pass> <N> [suffix] >> out

Units: rocprofv3 reports both counters in KB (x1024 bytes).  gfx950 correction: FETCH_SIZE
tallies the 128-byte requests of 16-byte-per-lane streaming loads at 64 bytes, so it is doubled
(all loads of the PCG kernels are 16 B per lane); WRITE_SIZE is exact for such stores."""
import csv
import glob
import json
import os
import sys

KERNELS = {"k_update_xr": "k_update_xr", "k_curvature": "k_curvature", "k_update_p": "k_update_p",
           "k_pack": "k_pack", "k_unpack_tangent": "k_unpack_tangent", "k_init<": "k_init"}


def collect(directory, counter):
    sums, counts = {}, {}
    for path in glob.glob(os.path.join(directory, "**", "*counter_collection.csv"), recursive=True):
        with open(path, newline="") as f:
            for row in csv.DictReader(f):
                if row.get("Counter_Name") != counter:
                    continue
                name = row.get("Kernel_Name", "")
                for pat, key in KERNELS.items():
                    if pat in name:
                        sums[key] = sums.get(key, 0.0) + float(row["Counter_Value"])
                        counts[key] = counts.get(key, 0) + 1
                        break
    return {k: sums[k] / counts[k] for k in sums}, counts


def main():
    fetch_dir, write_dir, n = sys.argv[1], sys.argv[2], int(sys.argv[3])
    suffix = sys.argv[4] if len(sys.argv) > 4 else ""
    fetch, nf = collect(fetch_dir, "FETCH_SIZE")
    write, nw = collect(write_dir, "WRITE_SIZE")
    out = {"raw_KB": {}, "launches": {}}
    for k in sorted(set(fetch) | set(write)):
        f, w = fetch.get(k), write.get(k)
        out["raw_KB"][k] = {"FETCH_SIZE": f, "WRITE_SIZE": w}
        out["launches"][k] = {"FETCH_SIZE": nf.get(k, 0), "WRITE_SIZE": nw.get(k, 0)}
        if f is not None and w is not None:
            out[f"{k}_{n}{suffix}"] = (2.0 * f + w) * 1024.0
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()

"""Join the per-launch algorithmic bytes / flops of scripts/engine_product_driver.py with rocprofv3 passes
of the same program:

    python scripts/pmc_engine_table.py <launches.json> <kernel-trace dir> <FETCH_SIZE dir> <WRITE_SIZE dir> <SQ dir>

Dispatches are matched BY ORDER: the driver's last ``products x launches_per_product`` dispatches are its
products, every product issues the same launches in the same order.  Output: JSON (stdout) with one row
per kernel name -- launches per product, mean duration, algorithmic vs counted bytes, achieved GB/s on
the algorithmic bytes, TFLOP/s, MFMA-busy fraction, LDS bank-conflict fraction, waves.

Units / corrections (MI355X_MICROARCH.md, rocprofv3 PMC section): FETCH_SIZE and WRITE_SIZE are in KB;
on gfx950 FETCH_SIZE tallies the 128-byte requests of 16-byte-per-lane loads at 64 bytes -> doubled
(every load of these kernels is a global_load_dwordx4 or narrower; for the narrower ones the doubling
over-counts, so "counted" is an upper bound there).  SQ_BUSY_CYCLES is summed over the chip's shader
engines; SQ_VALU_MFMA_BUSY_CYCLES counts cycles per SIMD with an MFMA in flight: the busy fraction below
is MFMA-busy cycles / (4 SIMDs x 256 CUs x kernel duration x clock)."""
import csv
import glob
import json
import os
import sys

CLOCK_GHZ = 2.4  # MI355X peak engine clock; the fraction is a lower bound when the chip clocks lower


def dispatches(directory, pattern):
    rows = []
    for path in glob.glob(os.path.join(directory, "**", pattern), recursive=True):
        with open(path, newline="") as f:
            rows += list(csv.DictReader(f))
    return rows


def by_dispatch(rows):
    out = {}
    for r in rows:
        d = int(r["Dispatch_Id"])
        e = out.setdefault(d, {"name": r["Kernel_Name"]})
        e[r["Counter_Name"]] = e.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    return [out[k] for k in sorted(out)]


def tail_products(seq, launches, products):
    need = launches * products
    if len(seq) < need:
        raise SystemExit(f"only {len(seq)} dispatches, need {need}")
    return seq[len(seq) - need:]


def base(kernel):
    return kernel.split("<")[0].split("(")[0]


def align(seq, logged, products, name_of):
    """The rows of ``seq`` (dispatch order) that are the logged launches of the LAST ``products`` products: every
    product starts at the anchor kernel (its first logged launch); inside a product the logged launches are matched
    in order by kernel name, rows in between that match nothing (launches the driver has no cost model for) are
    skipped and counted."""
    anchor = base(logged[0]["kernel"])
    starts = [i for i, r in enumerate(seq) if anchor in name_of(r)]
    if len(starts) < products:
        raise SystemExit(f"only {len(starts)} products in the trace, need {products}")
    starts = starts[-products:] + [len(seq)]
    out, skipped = [], 0
    for p in range(products):
        i = starts[p]
        for want in logged:
            while i < starts[p + 1] and base(want["kernel"]) not in name_of(seq[i]):
                i, skipped = i + 1, skipped + 1
            if i >= starts[p + 1]:
                raise SystemExit(f"product {p}: logged launch {want['kernel']} not found in the trace")
            out.append(seq[i])
            i += 1
    return out, skipped


def main():
    meta = json.load(open(sys.argv[1]))
    trace_dir, fetch_dir, write_dir, sq_dir = sys.argv[2:6]
    L, P, launches = meta["launches_per_product"], meta["products"], meta["launches"]
    # (only the library's own kernels -- anonymous namespace, k_* -- are in the driver's log; a classifier head the
    # engine does not fuse adds rocBLAS / ATen launches to a product, which are skipped here)
    ours = "(anonymous namespace)::k_"
    trace = [r for r in dispatches(trace_dir, "*kernel_trace.csv") if ours in r["Kernel_Name"]]
    trace.sort(key=lambda r: int(r["Start_Timestamp"]))
    trace, skipped = align(trace, launches, P, lambda r: r["Kernel_Name"])
    passes = {}
    for key, d in (("FETCH_SIZE", fetch_dir), ("WRITE_SIZE", write_dir), ("SQ", sq_dir)):
        rows = [r for r in dispatches(d, "*counter_collection.csv") if ours in r["Kernel_Name"]]
        passes[key], _ = align(by_dispatch(rows), launches, P, lambda r: r["name"])
    table = {}
    for i in range(L * P):
        want = launches[i % L]
        name = trace[i]["Kernel_Name"]
        if want["kernel"].split("<")[0].split("(")[0] not in name:
            raise SystemExit(f"dispatch {i}: expected {want['kernel']}, trace has {name}")
        row = table.setdefault(want["kernel"], {"launches": 0, "us": 0.0, "alg_read": 0.0, "alg_written": 0.0,
                                                "flops": 0.0, "fetch": 0.0, "write": 0.0, "sq": {}})
        row["launches"] += 1
        row["us"] += (int(trace[i]["End_Timestamp"]) - int(trace[i]["Start_Timestamp"])) * 1e-3
        row["alg_read"] += want["read"]
        row["alg_written"] += want["written"]
        row["flops"] += want["flops"]
        row["fetch"] += 2.0 * 1024.0 * passes["FETCH_SIZE"][i].get("FETCH_SIZE", 0.0)
        row["write"] += 1024.0 * passes["WRITE_SIZE"][i].get("WRITE_SIZE", 0.0)
        for k, v in passes["SQ"][i].items():
            if k != "name":
                row["sq"][k] = row["sq"].get(k, 0.0) + v
    out = {}
    for name, r in table.items():
        n = r["launches"]
        us = r["us"] / n
        alg = (r["alg_read"] + r["alg_written"]) / n
        sq = {k: v / n for k, v in r["sq"].items()}
        e = {
            "launches_per_product": n // P,
            "avg_us": round(us, 2),
            "alg_MB": round(alg / 1e6, 3),
            "counted_MB": round((r["fetch"] + r["write"]) / n / 1e6, 3),
            "counted_over_alg": round((r["fetch"] + r["write"]) / max(1.0, r["alg_read"] + r["alg_written"]), 2),
            "GBps_on_alg_bytes": round(alg / us / 1e3, 1),
            "frac_of_8TBps": round(alg / us / 1e3 / 8000.0, 3),
            "TFLOPs": round(r["flops"] / n / us / 1e6, 2),
            "waves": round(sq.get("SQ_WAVES", 0.0)),
        }
        mfma = sq.get("SQ_VALU_MFMA_BUSY_CYCLES")
        if mfma is not None:
            e["mfma_busy_frac"] = round(mfma / (4 * 256 * us * 1e-6 * CLOCK_GHZ * 1e9), 4)
        if sq.get("SQ_INSTS_VALU_MFMA_MOPS") is not None:
            e["mfma_mops"] = round(sq["SQ_INSTS_VALU_MFMA_MOPS"])
        if sq.get("SQ_LDS_IDX_ACTIVE"):
            e["lds_conflict_frac"] = round(sq.get("SQ_LDS_BANK_CONFLICT", 0.0) / sq["SQ_LDS_IDX_ACTIVE"], 3)
        if sq.get("SQ_WAVE_CYCLES"):
            e["wave_wait_frac"] = round(sq.get("SQ_WAIT_ANY", 0.0) / sq["SQ_WAVE_CYCLES"], 3)
        if sq.get("SQ_BUSY_CYCLES") is not None:
            e["sq_busy_cycles"] = round(sq["SQ_BUSY_CYCLES"])
        out[name] = e
    total = sum(v["avg_us"] * v["launches_per_product"] for v in out.values())
    print(json.dumps({"per_kernel": out, "sum_us_per_product": round(total, 1), "products": P,
                      "launches_per_product": L, "own_launches_without_cost_model_per_product": skipped / P}, indent=1))


if __name__ == "__main__":
    main()

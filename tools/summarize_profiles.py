#!/usr/bin/env python3
"""Turn rocprofv3 outputs under gpurun_out/ into the small committed summaries under profiles/.

    python tools/summarize_profiles.py r01

Inputs (written on the GPU box, see profiles/README.md for the exact commands):
  gpurun_out/prof_stats/<tag>_kernel_stats.csv        rocprofv3 --kernel-trace --stats   -- python bench.py ...
  gpurun_out/prof_fetch/<tag>_counter_collection.csv  rocprofv3 --kernel-trace --pmc FETCH_SIZE (own pass)
  gpurun_out/prof_write/<tag>_counter_collection.csv  rocprofv3 --kernel-trace --pmc WRITE_SIZE (own pass)
HBM traffic follows MI355X_MICROARCH.md (HBM section): bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 -- on gfx950
FETCH_SIZE reports half the bytes of wide coalesced reads (checked here on gn_apply, which reads exactly what it
writes: its raw FETCH_SIZE is half its WRITE_SIZE).
"""
import collections
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "gpurun_out")
P = os.path.join(ROOT, "profiles")


def agg(path, ctr):
    d = collections.defaultdict(lambda: [0, 0.0])
    with open(path) as f:
        for r in csv.DictReader(f):
            if r["Counter_Name"] == ctr:
                d[r["Kernel_Name"]][0] += 1
                d[r["Kernel_Name"]][1] += float(r["Counter_Value"])
    return d


def main(tag):
    os.makedirs(P, exist_ok=True)
    rows = list(csv.DictReader(open(os.path.join(G, "prof_stats", f"{tag}_kernel_stats.csv"))))
    with open(os.path.join(P, f"{tag}_kernel_stats.csv"), "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
        for r in rows:
            if float(r["Percentage"]) >= 0.005:
                w.writerow([r["Name"][:120], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"],
                            r["MinNs"], r["MaxNs"]])
    fetch = agg(os.path.join(G, "prof_fetch", f"{tag}_counter_collection.csv"), "FETCH_SIZE")
    write = agg(os.path.join(G, "prof_write", f"{tag}_counter_collection.csv"), "WRITE_SIZE")
    out = {}
    for k in fetch:
        n, fs = fetch[k]
        ws = write.get(k, [0, 0.0])[1]
        if fs + ws < 1024:
            continue
        out[k[:120]] = {"launches": n, "FETCH_SIZE_KB_per_launch_raw": fs / n,
                        "WRITE_SIZE_KB_per_launch": ws / n,
                        "hbm_bytes_per_launch": (2.0 * fs + ws) * 1024.0 / n}
    dom = ([k for k in out if "conv3x3_f43_kernel<2>" in k] or [k for k in out if "conv3x3_wino_kernel<2>" in k] or
           [k for k in out if "conv3x3_halo_kernel<2, 2, 2, 2, 2>" in k])
    summary = {"unit_note": "hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024 (gfx950 FETCH_SIZE halving correction)",
               "dominant_kernel": dom[0] if dom else None, "kernels": out}
    with open(os.path.join(P, f"{tag}_hbm_traffic.json"), "w") as f:
        json.dump(summary, f, indent=1)
    print("wrote", os.path.join(P, f"{tag}_kernel_stats.csv"), os.path.join(P, f"{tag}_hbm_traffic.json"))
    for k in dom:
        print(k, out[k])


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "r01")

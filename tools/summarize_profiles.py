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
    for sub, name in (("prof_bf16", "bf16"), ("prof_cfg5", "config5_rk4_fp16")):     # other modes: --stats summaries only
        src = os.path.join(G, sub, f"{tag}_kernel_stats.csv")
        if os.path.exists(src):
            with open(os.path.join(P, f"{tag}_{name}_kernel_stats.csv"), "w", newline="") as f:
                w = csv.writer(f)
                w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
                for r in csv.DictReader(open(src)):
                    if float(r["Percentage"]) >= 0.005:
                        w.writerow([r["Name"][:120], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"],
                                    r["MinNs"], r["MaxNs"]])
            print("wrote", os.path.join(P, f"{tag}_{name}_kernel_stats.csv"))
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
    dom = ([k for k in out if "conv3x3_w2d_kernel<2, 2>" in k] or [k for k in out if "conv3x3_w2d_kernel<2>" in k] or [k for k in out if "conv3x3_f43_kernel<2, false, 2>" in k] or [k for k in out if "conv3x3_f43_kernel<2, false>" in k] or [k for k in out if "conv3x3_f43_kernel<2>" in k] or
           [k for k in out if "conv3x3_wino_kernel<2>" in k] or
           [k for k in out if "conv3x3_halo_kernel<2, 2, 2, 2, 2>" in k])
    summary = {"unit_note": "hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024 (gfx950 FETCH_SIZE halving correction)",
               "dominant_kernel": dom[0] if dom else None, "kernels": out}
    # the launches bench.py brackets as the dominant kernel (model_plan.hip: unsplit 3x3 convs with fused GroupNorm
    # input on the Winograd kernels) run under more than one instantiation; their launch-weighted mean is what
    # compares with roofline.algorithmic_bytes_per_launch_avg
    dset = [k for k in out if "conv3x3_w2d_kernel<2, " in k or "conv3x3_f43_kernel<2, false, 2>" in k]
    if dset:
        n = sum(out[k]["launches"] for k in dset)
        summary["dominant_set"] = {"kernels": dset, "launches": n,
                                   "hbm_bytes_per_launch": sum(out[k]["launches"] * out[k]["hbm_bytes_per_launch"]
                                                               for k in dset) / n}
    with open(os.path.join(P, f"{tag}_hbm_traffic.json"), "w") as f:
        json.dump(summary, f, indent=1)
    print("wrote", os.path.join(P, f"{tag}_kernel_stats.csv"), os.path.join(P, f"{tag}_hbm_traffic.json"))
    sq_path = os.path.join(G, "prof_sq", f"{tag}_counter_collection.csv")
    if dom and os.path.exists(sq_path):                     # SQ counters of the dominant kernel (own pass)
        vals, durs = collections.defaultdict(list), []
        with open(sq_path) as f:
            for r in csv.DictReader(f):
                if r["Kernel_Name"][:120] == dom[0]:
                    vals[r["Counter_Name"]].append(float(r["Counter_Value"]))
                    if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
                        durs.append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
        m = {k: sum(v) / len(v) for k, v in vals.items()}
        dur = sum(durs) / len(durs)
        cyc = m["GRBM_GUI_ACTIVE"] / 8.0
        with open(os.path.join(P, f"{tag}_pmc_dominant_kernel.txt"), "w") as f:
            f.write("# rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY "
                    "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE -- python bench.py --steps 1 --warmup 1 "
                    "--no-cpu-baseline --no-alt\n")
            f.write(f"# kernel: {dom[0]}, {len(durs)} launches (warm-up, capture, timed and event-bracketed sampler passes at [8,1,256,256], N=5), "
                    f"avg {dur / 1e3:.1f} us (profiled run)\n")
            for k in sorted(m):
                f.write(f"{k:28s} per launch {m[k]:.4e}\n")
            f.write(f"effective clock (GRBM_GUI_ACTIVE / 8 XCD / duration): {cyc / dur:.2f} GHz\n")
            f.write("MFMA busy fraction per SIMD (SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMD x cycles)): "
                    f"{m['SQ_VALU_MFMA_BUSY_CYCLES'] / (1024 * cyc):.3f}\n")
            f.write(f"fp32 matrix peak at that clock: {64 * 1024 * cyc / dur / 1e3:.1f} TFLOP/s "
                    "(64 FLOP/clk/SIMD x 1024 SIMD)\n")
            if "SQ_BUSY_CU_CYCLES" in m:
                cu = m["SQ_BUSY_CU_CYCLES"] / 256.0
                f.write(f"shader clock from the CUs' own busy cycles (SQ_BUSY_CU_CYCLES / 256 CUs / duration): {cu / dur:.2f} GHz; "
                        f"MFMA busy fraction per SIMD at that clock: {m['SQ_VALU_MFMA_BUSY_CYCLES'] / (1024 * cu):.3f}\n")
        print("wrote", os.path.join(P, f"{tag}_pmc_dominant_kernel.txt"))
    sq16 = os.path.join(G, "prof_sq16", f"{tag}_counter_collection.csv")
    if os.path.exists(sq16):                                # SQ counters of the 16-bit producer / consumer conv (bf16 run)
        # PER INSTANTIATION (round 5 blended <2, false, 2> with <2, false, 1>: 128- and 64-channel blocks are different kernels)
        vals, durs = collections.defaultdict(lambda: collections.defaultdict(list)), collections.defaultdict(list)
        with open(sq16) as f:
            for r in csv.DictReader(f):
                if "conv3x3_pc16_kernel<" in r["Kernel_Name"]:
                    name = r["Kernel_Name"][:120]
                    vals[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
                    if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
                        durs[name].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
        if durs:
            with open(os.path.join(P, f"{tag}_pmc_bf16_kernel.txt"), "w") as f:
                f.write("# rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY "
                        "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE -- python bench.py --steps 1 --warmup 1 "
                        "--precision bf16 --no-cpu-baseline --no-alt\n"
                        "# one block per template instantiation <GN, F16, NJ> (NJ = 2: 128-channel blocks, 1: 64-channel blocks)\n")
                for name in sorted(durs, key=lambda k: -sum(durs[k])):
                    m = {k: sum(v) / len(v) for k, v in vals[name].items()}
                    dur = sum(durs[name]) / len(durs[name])
                    cyc = m["GRBM_GUI_ACTIVE"] / 8.0
                    f.write(f"\n# kernel: {name}, {len(durs[name])} launches (its shapes at [8,1,256,256], N=5), avg {dur / 1e3:.1f} us "
                            "(profiled run)\n")
                    for k in sorted(m):
                        f.write(f"{k:28s} per launch {m[k]:.4e}\n")
                    f.write(f"effective clock (GRBM_GUI_ACTIVE / 8 XCD / duration): {cyc / dur:.2f} GHz\n")
                    f.write("MFMA busy fraction per SIMD (SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMD x cycles)): "
                            f"{m['SQ_VALU_MFMA_BUSY_CYCLES'] / (1024 * cyc):.3f}\n")
                    if "SQ_BUSY_CU_CYCLES" in m:
                        cu = m["SQ_BUSY_CU_CYCLES"] / 256.0
                        f.write(f"shader clock from the CUs' own busy cycles (SQ_BUSY_CU_CYCLES / 256 CUs / duration): {cu / dur:.2f} GHz; "
                                f"MFMA busy fraction per SIMD at that clock: {m['SQ_VALU_MFMA_BUSY_CYCLES'] / (1024 * cu):.3f}\n")
                f.write("# (s_memtime inside the kernel agrees with the second clock: 1.3-1.7 GHz against the 100 MHz counter, "
                        "tools/pc16_ts.py -- the part clocks this kernel down; GRBM_GUI_ACTIVE does not follow it)\n")
            print("wrote", os.path.join(P, f"{tag}_pmc_bf16_kernel.txt"))
    for k in dom:
        print(k, out[k])


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "r01")

#!/usr/bin/env python3
"""Per-shape kernel times of tools/bench_wino.py from a rocprofv3 --kernel-trace csv (f43 vs w2d, min / median)."""
import csv
import itertools
import sys

rows = list(csv.DictReader(open(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/prof_w2d/w2d_kernel_trace.csv")))
seq = []
for r in rows:
    n = r["Kernel_Name"]
    if "conv3x3_f43_kernel" in n or "conv3x3_w2d_kernel" in n:
        seq.append(("w2d" if "w2d" in n else "f43" + n[n.index("<"):n.index(">") + 1],
                    (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
groups = [(k, [x[1] for x in g]) for k, g in itertools.groupby(seq, key=lambda t: t[0])]
for (k1, d1), (k2, d2) in zip(groups[0::2], groups[1::2]):
    m1, m2 = sorted(d1)[len(d1) // 2], sorted(d2)[len(d2) // 2]
    print(f"{k1:18s} min {min(d1):8.1f} med {m1:8.1f} us | {k2:4s} min {min(d2):8.1f} med {m2:8.1f} us | w2d/f43 (min) {min(d2)/min(d1):.3f}")

#!/usr/bin/env python3
"""Kernel-level timing of the 3x3 halo kernels (direct / F(2,3) / F(4,3), with and without the fused GroupNorm+SiLU
input stage) on the dominant shape; run under rocprofv3 --kernel-trace and read the per-kernel durations.

    rocprofv3 --kernel-trace -d gpurun_out/wino -o w --output-format csv -- python tools/bench_wino.py
    python tools/bench_wino.py --summarize gpurun_out/wino/w_kernel_trace.csv
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

B, H, W, Cout = 8, 256, 256, 128
CINS = (128, 256)


def summarize(path):
    import collections
    import csv
    d = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"]
        if "conv3x3" in k:
            d[(k[:48], r.get("Grid_Size") or r.get("Grid_Size_X", "?"))].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    for k, v in sorted(d.items()):
        print(f"{k[0]:50s} grid {k[1]:>8s} n={len(v):3d} min {min(v):8.1f} us  median {sorted(v)[len(v)//2]:8.1f} us")


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--summarize":
        summarize(sys.argv[2])
        sys.exit(0)
    import torch
    import _gpu as G
    for Cin in CINS:
        g = torch.Generator().manual_seed(0)
        x = torch.randn(B, Cin, H, W, generator=g)
        w = torch.randn(Cout, Cin, 3, 3, generator=g) / (Cin * 9) ** 0.5
        gam, bet = torch.ones(Cin), torch.zeros(Cin)
        for _ in range(3):
            G.conv2d(x, w)
            G.conv3x3_f23(x, w, form="f23")
            G.conv3x3_f23(x, w, form="f43")
            G.conv3x3_f23(x, w, gam, bet, form="f43")
        print("done", Cin, flush=True)

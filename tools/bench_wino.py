#!/usr/bin/env python3
"""K-length sweep of the Winograd / direct halo 3x3 kernels (runs on the GPU box): time = a + b * Cin.

    rocprofv3 --kernel-trace --stats -d gpurun_out/wino -o w --output-format csv -- python tools/bench_wino.py
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch

import _gpu as G

B, H, W, Cout = 8, 256, 256, 128
for Cin in (32, 64, 128, 256, 512):
    g = torch.Generator().manual_seed(0)
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / (Cin * 9) ** 0.5
    for _ in range(3):
        G.conv3x3_f23(x, w)
        G.conv2d(x, w)
    print("done", Cin, flush=True)

#!/usr/bin/env python3
"""Shape sweep of the 16-bit storage modes against the fp32 mode on the full network (synthetic weights): batch sizes and
frame counts that take different kernel paths (16x16-tile halo, per-tap halo, flat, split-K, odd tile counts).  Prints the
rel-L2 of every (B, T, mode); anything far from the ~5e-3 (bf16) / ~6e-4 (fp16) of the headline shape points at a
shape-dependent defect."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch

import _cases as C
from flowmse_amd.util import synth

if __name__ == "__main__":
    import test_gpu_model as TG
    full = TG._model(C.FULL)
    worst = {"bf16": 0.0, "fp16": 0.0}
    for B, T in ((1, 64), (1, 192), (1, 256), (2, 320), (3, 128), (5, 64), (8, 256), (16, 128), (4, 512)):
        y = C.c64(synth.synth_spectrogram(B, B, 256, T)).cuda()
        x = C.c64(synth.complex_normal(3 + T, 1, (B, 1, 256, T), 0.5)).cuda()
        t = torch.linspace(0.1, 0.9, B, device="cuda")
        full.dnn.set_precision("fp32")
        ref = full(x, t, y).clone()
        for mode in ("bf16", "fp16"):
            full.dnn.set_precision(mode)
            got = full(x, t, y)
            again = full(x, t, y)
            err = C.rel_l2(got.cpu(), ref.cpu())
            worst[mode] = max(worst[mode], err)
            print(f"B={B:2d} T={T:4d} {mode}: rel-L2 vs fp32 mode {err:.3e}  finite={bool(torch.isfinite(torch.view_as_real(got)).all())} "
                  f"deterministic={bool(torch.equal(got, again))}", flush=True)
        full.dnn.set_precision("fp32")
    print("worst", worst)
    assert worst["bf16"] < 2e-2 and worst["fp16"] < 3e-3

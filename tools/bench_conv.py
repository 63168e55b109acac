#!/usr/bin/env python3
"""Micro-benchmark of the conv kernel on the network's dominant shapes (runs on the GPU box).

    python tools/bench_conv.py [--check]
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from flowmse_amd import _lib

L = _lib.lib
SHAPES = [  # B, H, W, C1, C2, Cout, taps
    (8, 256, 256, 128, 0, 128, 9),
    (8, 256, 256, 128, 128, 128, 9),
    (8, 128, 128, 128, 0, 128, 9),
    (8, 64, 64, 256, 0, 256, 9),
    (8, 64, 64, 256, 256, 256, 9),
    (8, 32, 32, 256, 0, 256, 9),
    (8, 16, 16, 256, 0, 256, 9),
    (8, 8, 8, 256, 256, 256, 9),
    (8, 4, 4, 256, 256, 256, 9),
    (8, 4, 4, 256, 0, 256, 9),
    (8, 256, 256, 128, 128, 128, 1),
    (8, 256, 256, 128, 0, 4, 9),
]


def run(shape, iters=5, check=False):
    B, H, W, C1, C2, Cout, taps = shape
    g = torch.Generator(device="cpu").manual_seed(0)
    a1 = torch.randn(B, H, W, C1, generator=g).cuda()
    a2 = torch.randn(B, H, W, C2, generator=g).cuda() if C2 else None
    w = (torch.randn(Cout, taps, C1 + C2, generator=g) / ((C1 + C2) * taps) ** 0.5).cuda()
    bias = torch.randn(Cout, generator=g).cuda()
    res = torch.randn(B, H, W, Cout, generator=g).cuda()
    out = torch.empty(B, H, W, Cout, device="cuda")
    st = _lib.current_stream()
    nscr = L.flowse_op_conv2d_scratch_floats(B, H, W, C1 + C2, Cout, taps)
    scratch = torch.empty(max(nscr, 1), device="cuda")

    def call():
        _lib.check(L.flowse_op_conv2d(_lib.ptr(a1), C1, _lib.ptr(a2), C2, _lib.ptr(w), _lib.ptr(bias), None, 0,
                                      _lib.ptr(res), _lib.ptr(out), B, H, W, Cout, taps, 0.7071, _lib.ptr(scratch), st))
    call()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        call()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    flops = 2.0 * B * H * W * Cout * taps * (C1 + C2)
    line = f"{str(shape):44s} {ms:8.3f} ms  {flops/ms/1e9:7.1f} TF/s"
    if check:
        k = 3 if taps == 9 else 1
        x = torch.cat([a1, a2], 3) if C2 else a1
        wt = w.reshape(Cout, k, k, C1 + C2).permute(0, 3, 1, 2).contiguous()
        ref = torch.nn.functional.conv2d(x.permute(0, 3, 1, 2), wt, bias, padding=k // 2)
        ref = (ref + res.permute(0, 3, 1, 2)) * 0.7071
        err = float((out.permute(0, 3, 1, 2) - ref).norm() / ref.norm())
        line += f"  rel-L2 vs torch(gpu) {err:.2e}"
    print(line, flush=True)


if __name__ == "__main__":
    for s in SHAPES:
        run(s, check="--check" in sys.argv)

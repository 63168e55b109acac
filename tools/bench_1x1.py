#!/usr/bin/env python3
"""fp32 1x1 convs of the network's large levels through flowse_op_conv2d: streaming kernel (with the weight-copy scratch)
vs the flat kernel (without).  Run under rocprofv3 --kernel-trace --stats for kernel-only times."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from flowmse_amd import _lib

L = _lib.lib
SHAPES = [(8, 256, 256, 128, 128, 128), (8, 256, 256, 128, 0, 128), (8, 128, 128, 128, 128, 128), (8, 128, 128, 256, 128, 128),
          (8, 128, 128, 256, 0, 256), (8, 64, 64, 256, 256, 256), (8, 64, 64, 256, 128, 256)]
for (B, H, W, C1, C2, Cout) in SHAPES:
    g = torch.Generator(device="cpu").manual_seed(0)
    a1 = torch.randn(B, H, W, C1, generator=g).cuda()
    a2 = torch.randn(B, H, W, C2, generator=g).cuda() if C2 else None
    w = (torch.randn(Cout, 1, C1 + C2, generator=g) / (C1 + C2) ** 0.5).cuda()
    bias = torch.randn(Cout, generator=g).cuda()
    st = _lib.current_stream()
    line = f"{str((B, H, W, C1, C2, Cout)):36s}"
    outs = []
    for mode in ("stream", "flat"):
        out = torch.empty(B, H, W, Cout, device="cuda")
        n = L.flowse_op_conv2d_scratch_floats(B, H, W, C1 + C2, Cout, 1) if mode == "stream" else 0
        scratch = torch.empty(max(n, 1), device="cuda") if mode == "stream" else None

        def call():
            _lib.check(L.flowse_op_conv2d(_lib.ptr(a1), C1, _lib.ptr(a2), C2, _lib.ptr(w), _lib.ptr(bias), None, 0, None,
                                          _lib.ptr(out), B, H, W, Cout, 1, 1.0, _lib.ptr(scratch), st))
        call()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(8):
            call()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 8
        line += f"  {mode}: {ms*1e3:7.1f} us {2.0*B*H*W*Cout*(C1+C2)/ms/1e9:6.1f} TF/s"
        outs.append(out)
    line += f"   rel {float((outs[0]-outs[1]).norm()/outs[1].norm()):.1e}"
    print(line, flush=True)

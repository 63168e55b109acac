#!/usr/bin/env python3
"""Run the dominant kernel (F(4,3) 3x3 conv with fused GroupNorm+SiLU input) alone on its two big shapes:
target for rocprofv3 --pmc passes (tools/pmc_f43.sh) and for A/B timing of kernel variants.

    python tools/f43_probe.py [iters]
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch

SHAPES = [(8, 256, 256, 128, 0, 128), (8, 256, 256, 128, 128, 128), (8, 128, 128, 128, 0, 128), (8, 64, 64, 256, 0, 256)]

if __name__ == "__main__":
    from flowmse_amd import _lib
    L = _lib.lib
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 5
    only = int(os.environ.get("PROBE_SHAPE", "-1"))
    for si, (B, H, W, C1, C2, Cout) in enumerate(SHAPES):
        if only >= 0 and si != only:
            continue
        g = torch.Generator().manual_seed(0)
        C = C1 + C2
        a1 = torch.randn(B, H, W, C1, generator=g).cuda()
        a2 = torch.randn(B, H, W, C2, generator=g).cuda() if C2 else None
        w = (torch.randn(Cout, 9, C, generator=g) / (C * 9) ** 0.5).cuda()
        bias = torch.randn(Cout, generator=g).cuda()
        res = torch.randn(B, H, W, Cout, generator=g).cuda()
        gam, bet = (1 + 0.1 * torch.randn(C, generator=g)).cuda(), (0.1 * torch.randn(C, generator=g)).cuda()
        out = torch.empty(B, H, W, Cout, device="cuda")
        scratch = torch.empty(L.flowse_op_conv3x3_f43_scratch_floats(B, H, W, C, Cout), device="cuda")
        st = _lib.current_stream()

        def call():
            _lib.check(L.flowse_op_conv3x3_f43(_lib.ptr(a1), C1, _lib.ptr(a2), C2, _lib.ptr(gam), _lib.ptr(bet), 1e-6, 1,
                                               _lib.ptr(w), _lib.ptr(bias), None, 0, _lib.ptr(res), _lib.ptr(out), B, H, W,
                                               Cout, 0.7071, _lib.ptr(scratch), st))
        call()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        # the op also runs gn_stats + gn_finalize + the weight transform: time a batch and report per call
        e0.record()
        for _ in range(iters):
            call()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / iters
        flops = 2.0 * B * H * W * Cout * 9 * C
        print(f"shape {si} {(B, H, W, C1, C2, Cout)}: {ms:.3f} ms per op call (incl. gn_stats/finalize/weight xform) "
              f"-> >= {flops / ms / 1e9:.1f} TF/s algorithmic", flush=True)

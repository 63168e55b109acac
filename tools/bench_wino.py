#!/usr/bin/env python3
"""A-B of the two fp32 Winograd forms of the fused 3x3 conv on the network's large shapes (runs on the GPU box):
1-D F(4,3) (conv3x3_f43_kernel) vs 2-D F(4,3) x F(2,3) (conv3x3_w2d_kernel), GroupNorm + SiLU input, residual, per-sample
bias -- the ResnetBlock call.  The GroupNorm statistics launches of the op-level entry are outside the timed region
(the timed calls run with gamma == NULL unless --gn; kernel-only timing: use rocprofv3 --kernel-trace --stats).

    python tools/bench_wino.py [--gn] [--iters N] [--only SHAPE_INDEX] [--form f43|w2d]
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from flowmse_amd import _lib

L = _lib.lib
SHAPES = [  # B, H, W, C1, C2, Cout   (launches per NFE at [8,1,256,256])
    (8, 256, 256, 128, 0, 128),      # conv1 @256: 6, conv0 @256: 2+1
    (8, 256, 256, 128, 128, 128),    # conv0 @256 up path: 3
    (8, 128, 128, 128, 0, 128),
    (8, 128, 128, 128, 128, 128),
    (8, 128, 128, 256, 0, 256),
    (8, 128, 128, 256, 128, 128),
    (8, 64, 64, 256, 0, 256),
    (8, 64, 64, 256, 256, 256),
    (8, 32, 32, 256, 0, 256),
    (1, 256, 256, 128, 0, 128),
    (1, 128, 128, 128, 0, 128),
    (8, 64, 48, 256, 0, 256),        # 384 blocks of 64 channels (T = 192 at the 64-row level)
    (8, 64, 32, 256, 0, 256),        # 256
    (8, 32, 48, 256, 0, 256),        # 192 (T = 384 at the 32-row level)
    (8, 128, 48, 128, 0, 128),       # 384 (T = 96 ... the 128-row level of short utterances)
]


def run(shape, iters, gn, forms=("f43", "w2d")):
    B, H, W, C1, C2, Cout = shape
    C = C1 + C2
    g = torch.Generator(device="cpu").manual_seed(0)
    a1 = torch.randn(B, H, W, C1, generator=g).cuda()
    a2 = torch.randn(B, H, W, C2, generator=g).cuda() if C2 else None
    w = (torch.randn(Cout, 9, C, generator=g) / (C * 9) ** 0.5).cuda()
    bias = torch.randn(Cout, generator=g).cuda()
    gamma = (1 + 0.1 * torch.randn(C, generator=g)).cuda()
    beta = (0.1 * torch.randn(C, generator=g)).cuda()
    res = torch.randn(B, H, W, Cout, generator=g).cuda()
    st = _lib.current_stream()
    outs, line = {}, f"{str(shape):34s}"
    for form in forms:
        out = torch.empty(B, H, W, Cout, device="cuda")
        scratch = torch.empty(getattr(L, f"flowse_op_conv3x3_{form}_scratch_floats")(B, H, W, C, Cout), device="cuda")
        fn = getattr(L, f"flowse_op_conv3x3_{form}")

        def call():
            rc = fn(_lib.ptr(a1), C1, _lib.ptr(a2), C2, _lib.ptr(gamma) if gn else None, _lib.ptr(beta) if gn else None, 1e-6, 1,
                    _lib.ptr(w), _lib.ptr(bias), None, 0, _lib.ptr(res), _lib.ptr(out), B, H, W, Cout, 0.7071,
                    _lib.ptr(scratch), st)
            return rc
        if call() != 0:
            line += f"  {form}: not covered"
            continue
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            call()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / iters
        flops = 2.0 * B * H * W * Cout * 9 * C
        line += f"  {form}: {ms*1e3:8.1f} us {flops/ms/1e9:6.1f} TF/s(alg)"
        outs[form] = out
    if len(outs) == 2:
        d = float((outs["w2d"] - outs["f43"]).norm() / outs["f43"].norm())
        line += f"   w2d vs f43 rel-L2 {d:.2e}"
    print(line, flush=True)


if __name__ == "__main__":
    it = int(sys.argv[sys.argv.index("--iters") + 1]) if "--iters" in sys.argv else 10
    print("# per call incl. the weight transform" + (" and GroupNorm statistics launches" if "--gn" in sys.argv else "") +
          " (op-level entry); kernel-only times: rocprofv3 --kernel-trace --stats")
    only = int(sys.argv[sys.argv.index("--only") + 1]) if "--only" in sys.argv else None
    forms = (sys.argv[sys.argv.index("--form") + 1],) if "--form" in sys.argv else ("f43", "w2d")
    if "--shape" in sys.argv:                                  # B,H,W,C1,C2,Cout
        SHAPES = [tuple(int(v) for v in sys.argv[sys.argv.index("--shape") + 1].split(","))]
    for k, s in enumerate(SHAPES):
        if only is None or k == only:
            run(s, it, "--gn" in sys.argv, forms)

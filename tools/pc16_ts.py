#!/usr/bin/env python3
"""Where the cycles of conv3x3_pc16_kernel go (measurement build only).

    python tools/build_variants.py meas "-DFLOWSE_MEASURE"
    FLOWSE_LIB_PATH=flowmse_amd/variants/meas/libflowse_hip.so python tools/pc16_ts.py [B C1 C2 Cout H W gn res silu]

Runs one 3x3 conv on 16-bit activations through flowse_op_conv2d_16 and prints, per role, the s_memtime accumulators
(median over blocks, cycles): consumers -- fragments + MFMA issue / step barrier / prologue / output stage; producers --
staging work / step barrier / prologue / statistics barrier."""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
from flowmse_amd import _lib


def main():
    a = [int(v) for v in sys.argv[1:]]
    B, C1, C2, Cout, H, W, gn, res, silu = (a + [8, 128, 0, 128, 256, 256, 1, 1, 1][len(a):])[:9]
    L = _lib.lib
    g = torch.Generator().manual_seed(1)
    x1 = torch.randn(B, H, W, C1, generator=g).cuda()
    x2 = torch.randn(B, H, W, C2, generator=g).cuda() if C2 else None
    Cin = C1 + C2
    w = (torch.randn(Cout, 9, Cin, generator=g) / (9 * Cin) ** 0.5).cuda()
    bias = torch.randn(Cout, generator=g).cuda()
    r = torch.randn(B, H, W, Cout, generator=g).cuda() if res else None
    mean = (0.2 * torch.randn(B, Cin, generator=g)).cuda() if gn else None
    scl = (1 + 0.2 * torch.randn(B, Cin, generator=g)).cuda() if gn else None
    beta = (0.2 * torch.randn(Cin, generator=g)).cuda() if gn else None
    out = torch.empty(B, H, W, Cout, device="cuda")
    scratch = torch.empty(3 << 30, dtype=torch.uint8, device="cuda")
    s = torch.cuda.current_stream().cuda_stream

    def run():
        _lib.check(L.flowse_op_conv2d_16(_lib.ptr(x1), C1, _lib.ptr(x2), C2, _lib.ptr(w), _lib.ptr(bias), _lib.ptr(r),
                                         _lib.ptr(mean), _lib.ptr(scl), _lib.ptr(beta), silu, _lib.ptr(out), B, H, W, Cout, 9,
                                         0.7071, 1, _lib.ptr(scratch), scratch.numel(), C.c_void_p(s)))
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        run()
    torch.cuda.synchronize()
    print(f"B={B} C={C1}+{C2}>{Cout} {H}x{W} gn={gn} silu={silu} res={res}: {(time.perf_counter() - t0) / 10 * 1e6:.1f} us per call "
          "(incl. the boundary conversions of flowse_op_conv2d_16)")
    raw = C.CDLL(os.environ.get("FLOWSE_LIB_PATH", os.path.join(ROOT, "flowmse_amd", "libflowse_hip.so")))
    if not hasattr(raw, "flowse_debug_pc_ts"):
        print("(library built without -DFLOWSE_MEASURE: no timestamps)")
        return
    buf = (C.c_ulonglong * (256 * 16))()
    raw.flowse_debug_pc_ts(buf, 256 * 16)
    t = np.frombuffer(buf, dtype=np.uint64).reshape(256, 16).astype(np.float64)
    names = ["cons: frag+mfma", "cons: chunk barrier", "", "cons: hand-off (round 6) + next tile setup", "", "cons: kernel entry -> loop end (ticks)",
             "cons: the same on the 100 MHz counter", "",
             "prod: requests", "prod: chunk barrier", "", "prod: drain of the hand-off tile", "prod: halo bursts",
             "prod: kernel entry -> loop end (ticks)", "prod: the same on the 100 MHz counter"]
    live = t[:, 5] > 0                                       # blocks that ran (a launch may have fewer than 256)
    for k, n in enumerate(names):
        if n:
            v = t[live, k]
            print(f"  {n:22s} median {np.median(v):10.0f}  min {v.min():10.0f}  max {v.max():10.0f} cycles")
    # start skew and per-block wall time (100 MHz counter -> us), by XCD (= blockIdx & 7)
    idx = np.nonzero(live)[0]
    t0 = t[live, 7] - t[live, 7].min()
    dur = t[live, 6]
    print(f"  blocks {live.sum()}: entry skew us  p50 {np.median(t0) / 100:.2f}  p90 {np.percentile(t0, 90) / 100:.2f}  max {t0.max() / 100:.2f};"
          f"  duration us  p10 {np.percentile(dur, 10) / 100:.2f}  p50 {np.median(dur) / 100:.2f}  p90 {np.percentile(dur, 90) / 100:.2f}  max {dur.max() / 100:.2f};"
          f"  last end {((t0 + dur).max()) / 100:.2f}")
    for x in range(8):
        m = (idx & 7) == x
        if m.any():
            print(f"    xcd {x}: n {m.sum():3d}  entry p50 {np.median(t0[m]) / 100:6.2f} max {t0[m].max() / 100:6.2f}   dur p50 {np.median(dur[m]) / 100:6.2f} max {dur[m].max() / 100:6.2f}")


if __name__ == "__main__":
    main()

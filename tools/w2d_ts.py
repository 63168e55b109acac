#!/usr/bin/env python3
"""Timeline of conv3x3_w2d_kernel blocks (measurement build: tools/build_variants.py w2dts -DFLOWSE_MEASURE_W2D).

    FLOWSE_LIB_PATH=flowmse_amd/variants/w2dts/libflowse_hip.so python tools/w2d_ts.py B,H,W,C1,C2,Cout

Stamps (100 MHz counter, wave 0 of blocks 0..255): 0 entry, 1 first halo staged, per slot s: 4+3s before the slot barrier,
5+3s after it, 6+3s end of the slot's last phase; 2/3 after an output stage."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from flowmse_amd import _lib

L = _lib.lib
B, H, W, C1, C2, Cout = [int(v) for v in sys.argv[1].split(",")]
Cc = C1 + C2
g = torch.Generator().manual_seed(0)
a1 = torch.randn(B, H, W, C1, generator=g).cuda()
a2 = torch.randn(B, H, W, C2, generator=g).cuda() if C2 else None
w = (torch.randn(Cout, 9, Cc, generator=g) / (Cc * 9) ** 0.5).cuda()
bias = torch.randn(Cout, generator=g).cuda()
gamma = (1 + 0.1 * torch.randn(Cc, generator=g)).cuda()
beta = (0.1 * torch.randn(Cc, generator=g)).cuda()
res = torch.randn(B, H, W, Cout, generator=g).cuda()
out = torch.empty(B, H, W, Cout, device="cuda")
scratch = torch.empty(L.flowse_op_conv3x3_w2d_scratch_floats(B, H, W, Cc, Cout), device="cuda")
st = _lib.current_stream()
for _ in range(3):
    _lib.check(L.flowse_op_conv3x3_w2d(_lib.ptr(a1), C1, _lib.ptr(a2), C2, _lib.ptr(gamma), _lib.ptr(beta), 1e-6, 1, _lib.ptr(w), _lib.ptr(bias),
                                       None, 0, _lib.ptr(res), _lib.ptr(out), B, H, W, Cout, 0.7071, _lib.ptr(scratch), st))
torch.cuda.synchronize()
raw = C.CDLL(os.environ["FLOWSE_LIB_PATH"])
buf = (C.c_ulonglong * (256 * 64))()
raw.flowse_debug_w2d_ts(buf, 256 * 64)
t = np.frombuffer(buf, dtype=np.uint64).reshape(256, 64).astype(np.float64)
live = t[:, 0] > 0
t0 = t[live, 0].min()
nch = Cc // 32
print(f"shape {sys.argv[1]}: {live.sum()} blocks stamped; entry skew p50 {np.median(t[live,0]-t0)/100:.2f} us max {(t[live,0]-t0).max()/100:.2f} us")
rel = (t[live] - t[live, :1]) / 100.0      # us since the block's own entry
print(f"  first halo staged (barrier passed)        p50 {np.median(rel[:,1]):6.2f} us")
k = 0
while 6 + 3 * k < 64 and np.median(t[live, 6 + 3 * k]) > 0:
    print(f"  slot {k:2d}: phases 0-2 done {np.median(rel[:,4+3*k]):6.2f}  barrier passed {np.median(rel[:,5+3*k]):6.2f}  phase 3 done {np.median(rel[:,6+3*k]):6.2f}"
          + (f"   <- tile ends; output stage done {np.median(rel[:,2+((k//nch)&1)]):6.2f}" if (k + 1) % nch == 0 else ""))
    k += 1
if os.environ.get("W2D_WAVES"):             # -DFLOWSE_MEASURE_W2D_WAVES build: [block][wave][slot] arrival at the slot barrier
    tw = t.reshape(256, 8, 8)
    lv = tw[:, 0, 0] > 0
    print("  per-wave arrival at the slot barrier, us behind the block's FIRST arriver (median over blocks); wave = (CH = w >> 2, h = w & 3)")
    for sl in range(8):
        if np.median(tw[lv, 0, sl]) <= 0:
            break
        arr = tw[lv, :, sl]
        d = (arr - arr.min(axis=1, keepdims=True)) / 100.0
        print(f"    slot {sl}: " + "  ".join(f"w{w}:{np.median(d[:, w]):5.2f}" for w in range(8)))

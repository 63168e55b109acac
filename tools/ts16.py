#!/usr/bin/env python3
"""Phase timestamps of the 16-bit halo kernel (library built with FLOWSE_BUILD_FLAGS=-DFLOWSE_TS): runs the bf16 bench
workload once in-process, reads the per-block s_memtime marks of the last conv1@256x256:128>128 launch and prints where
a block's time goes and how the blocks of one CU overlap.

    FLOWSE_BUILD_FLAGS=-DFLOWSE_TS python -m flowmse_amd.build && python tools/ts16.py
"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

if __name__ == "__main__":
    sys.argv = ["bench.py", "--steps", "1", "--warmup", "1", "--precision", os.environ.get("TS_PRECISION", "bf16"),
                "--no-cpu-baseline", "--no-alt"]     # TS_PRECISION=fp32: the F(4,3) kernel's marks (no per-tap marks there)
    os.environ["FLOWSE_NO_GRAPH"] = "1"
    import bench
    bench.main()
    from flowmse_amd import _lib
    nblk = int(os.environ.get("TS_BLOCKS", "4096"))
    buf = (C.c_ulonglong * (nblk * 10))()
    f = _lib.lib.flowse_debug_ts
    f.restype = C.c_int
    f.argtypes = [C.c_void_p, C.c_int]
    assert f(buf, nblk * 10) == 0
    t = np.frombuffer(buf, dtype=np.uint64).reshape(nblk, 10).astype(np.int64)
    t = t[t[:, 0] > 0]
    print("blocks with marks:", len(t))
    t0 = t[:, 0].min()
    span = t[:, 7].max() - t0
    print('launch span', span)
    ph = np.stack([t[:, 1] - t[:, 0], t[:, 6] - t[:, 1], t[:, 7] - t[:, 6], t[:, 7] - t[:, 0],
                   t[:, 2] / 6, t[:, 3] / 6, t[:, 4] / 6, t[:, 5] / 6, (t[:, 2] + t[:, 3] + t[:, 4] + t[:, 5]) / 6], 1)
    names = ["prologue", "mainloop", "epilogue", "total", "tap:rd+mfma", "tap:gn", "tap:w-wait+wr", "tap:barrier", "tap:total"]
    for k, n in enumerate(names):
        v = ph[:, k]
        print(f"{n:14s} mean {v.mean():9.0f}  p10 {np.percentile(v, 10):9.0f}  p50 {np.percentile(v, 50):9.0f}  p90 {np.percentile(v, 90):9.0f}")
    if os.environ.get("TS_PRECISION") == "fp32":        # F(4,3) kernel: slots 2..5 = epilogue marks (scatter / output stage per half)
        e = np.stack([t[:, 2] - t[:, 6], t[:, 3] - t[:, 2], t[:, 4] - t[:, 3], t[:, 7] - t[:, 4]], 1)
        for k, n in enumerate(["epi:scatter0", "epi:out0", "epi:scatter1", "epi:out1"]):
            print(f"{n:14s} mean {e[:, k].mean():9.0f}  p50 {np.percentile(e[:, k], 50):9.0f}")
    hw = t[:, 8]
    cu = (hw >> 8) & 0xF
    sh = (hw >> 12) & 0x1
    se = (hw >> 13) & 0x7
    xcc = t[:, 9] & 0xF
    key = ((xcc * 8 + se) * 2 + sh) * 16 + cu
    print("distinct CUs:", len(np.unique(key)))
    # one CU: list its blocks in start order
    for kk in np.unique(key)[:2]:
        rows = t[key == kk]
        rows = rows[np.argsort(rows[:, 0])]
        print(f"CU key {kk}: {len(rows)} blocks")
        for r in rows[:8]:
            print("   start %8d  pro %6d  main %6d  epi %6d  end %8d  blockIdx.x %d" % (r[0] - t0, r[1] - r[0], r[6] - r[1], r[7] - r[6], r[7] - t0, r[5]))
    busy = []
    for kk in np.unique(key):
        rows = t[key == kk]
        busy.append((rows[:, 7] - rows[:, 0]).sum() / max(1, rows[:, 7].max() - rows[:, 0].min()))
    print("mean concurrently-resident blocks per CU:", np.mean(busy))

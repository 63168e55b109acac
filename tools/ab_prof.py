#!/usr/bin/env python3
"""Per-op timing of one variant: python tools/ab_prof.py NAME PATTERN  -> total ms of ops whose label contains PATTERN"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for name in sys.argv[2:]:
    env = dict(os.environ)
    if name != "base":
        env["FLOWSE_LIB_PATH"] = os.path.join(ROOT, "flowmse_amd", "variants", name, "libflowse_hip.so")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "2", "--no-alt", "--no-cpu-baseline", "--profile-all"],
                       env=env, capture_output=True, text=True)
    tot, n = 0.0, 0
    rows = []
    for ln in r.stderr.splitlines():
        if ln.startswith("# ") and sys.argv[1] in ln and " n=" in ln:
            f = ln.split()
            ms = float(f[f.index("ms") - 1]); tot += ms
            rows.append((f[1], ms))
    val = [json.loads(l)["value"] for l in r.stdout.splitlines() if l.startswith("{")]
    print(f"{name:14s} {sys.argv[1]}: {tot:8.3f} ms per step   value {val[0] if val else None}   top: {rows[:3]}", flush=True)

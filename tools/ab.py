#!/usr/bin/env python3
"""A-B timing of library variants on the GPU box (built beforehand with tools/build_variants.py).

    python tools/ab.py [--bench-args "..."] NAME [NAME ...]      # NAME = variants/NAME, or "base" for the shipped build

Runs bench.py once per variant (fresh process, FLOWSE_LIB_PATH) and prints value / dominant-kernel time / fraction.
"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    a = sys.argv[1:]
    bargs = "--steps 5 --warmup 2 --no-alt --no-cpu-baseline"
    if a and a[0] == "--bench-args":
        bargs = a[1]
        a = a[2:]
    rows = []
    for spec in a:
        name, _, envs = spec.partition(":")              # NAME[:K=V,K=V]
        env = dict(os.environ)
        if name != "base":
            env["FLOWSE_LIB_PATH"] = os.path.join(ROOT, "flowmse_amd", "variants", name, "libflowse_hip.so")
        for kv in filter(None, envs.split(",")):
            k, _, v = kv.partition("=")
            env[k] = v
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + bargs.split(), env=env, capture_output=True,
                           text=True)
        line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        if r.returncode or not line:
            print(f"{spec:28s} FAILED rc={r.returncode}: {r.stderr[-400:]}", flush=True)
            continue
        d = json.loads(line[-1])
        ro = d.get("roofline", {})
        print(f"{spec:28s} value {d['value']:9.1f} frames/s  ms/step {d['ms_per_step']:8.3f}  dominant avg "
              f"{1e3 * ro.get('avg_launch_ms', 0):7.1f} us  frac {ro.get('frac', 0):.3f}  whole-path "
              f"{ro.get('whole_path_issued_frac', 0):.3f}  launches/nfe {d.get('launches_per_nfe')}", flush=True)
        rows.append((spec, d["value"]))
    return 0


if __name__ == "__main__":
    sys.exit(main())

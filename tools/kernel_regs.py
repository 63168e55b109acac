#!/usr/bin/env python3
"""Register / spill figures of every kernel of the library, from the compiler's own metadata (the numbers DESIGN.md quotes).

    python tools/kernel_regs.py [file.hip ...]      # default: every translation unit of flowmse_amd/build.py

Compiles each translation unit with the library's flags plus `-S` (device assembly, gfx950) into a temporary directory and
prints .vgpr_count / .agpr_count / .sgpr_count / spill counts / scratch bytes / LDS bytes per kernel.
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from flowmse_amd import build as fb


def demangle(names):
    try:
        out = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt"] + names, capture_output=True, text=True).stdout.split("\n")
        return [o for o in out if o] or names
    except Exception:
        return names


def main(files):
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    rows = []
    with tempfile.TemporaryDirectory() as tmp:
        for src in files:
            out = os.path.join(tmp, os.path.basename(src) + ".s")
            cmd = [hipcc] + fb.FLAGS + fb.EXTRA_FLAGS.get(os.path.basename(src), []) + ["--cuda-device-only", "-S",
                   os.path.join(fb.CSRC, src) if not os.path.isabs(src) else src, "-o", out]
            subprocess.check_call(cmd)
            text = open(out).read()
            for blk in text.split("  - .agpr_count:")[1:]:
                def f(key):
                    m = re.search(r"\.%s:\s+(\S+)" % key, blk)
                    return m.group(1) if m else "?"
                rows.append((os.path.basename(src), f("name"), "  " + blk.split("\n")[0].strip(), f("vgpr_count"), f("vgpr_spill_count"),
                             f("sgpr_count"), f("sgpr_spill_count"), f("private_segment_fixed_size"),
                             f("group_segment_fixed_size")))
    names = demangle([r[1] for r in rows])
    print(f"{'file':16s} {'vgpr':>4s} {'agpr':>4s} {'vspill':>6s} {'sgpr':>4s} {'sspill':>6s} {'scratch':>7s} {'lds':>6s}  kernel")
    for r, n in zip(rows, names):
        print(f"{r[0]:16s} {r[3]:>4s} {r[2].strip():>4s} {r[4]:>6s} {r[5]:>4s} {r[6]:>6s} {r[7]:>7s} {r[8]:>6s}  {n[:110]}")


if __name__ == "__main__":
    main(sys.argv[1:] or fb.SOURCES)

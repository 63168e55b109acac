#!/usr/bin/env python3
"""Build extra copies of libflowse_hip.so with different compile flags, for A-B timing on the GPU box.

    python tools/build_variants.py NAME "FLAGS" [NAME "FLAGS" ...]

Each variant lands in flowmse_amd/variants/NAME/libflowse_hip.so (git-ignored, travels with gpurun) and is selected
with FLOWSE_LIB_PATH=flowmse_amd/variants/NAME/libflowse_hip.so.  Objects are rebuilt only when the sources or flags
changed.
"""
import hashlib
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from flowmse_amd import build as fb


def build_variant(name, flags):
    out = os.path.join(ROOT, "flowmse_amd", "variants", name)
    os.makedirs(out, exist_ok=True)
    h = hashlib.sha256()
    for n in sorted(os.listdir(fb.CSRC)):
        if not n.endswith(".o"):
            h.update(open(os.path.join(fb.CSRC, n), "rb").read())
    h.update(flags.encode())
    lib, stamp = os.path.join(out, "libflowse_hip.so"), os.path.join(out, "stamp")
    if os.path.exists(lib) and os.path.exists(stamp) and open(stamp).read() == h.hexdigest():
        return lib
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    base = [f for f in fb.FLAGS]
    procs, objs = [], []
    for src in fb.SOURCES:
        obj = os.path.join(out, src.replace(".hip", ".o"))
        objs.append(obj)
        procs.append((src, subprocess.Popen([hipcc] + base + fb.EXTRA_FLAGS.get(src, []) + flags.split() + ["-c", os.path.join(fb.CSRC, src), "-o", obj],
                                            stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    for src, p in procs:
        o, _ = p.communicate()
        if p.returncode:
            raise RuntimeError(f"{name}: hipcc failed on {src}:\n{o.decode()}")
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs)
    open(stamp, "w").write(h.hexdigest())
    return lib


if __name__ == "__main__":
    a = sys.argv[1:]
    for i in range(0, len(a), 2):
        print(build_variant(a[i], a[i + 1]), flush=True)

"""Build libflowse_hip.so (gfx950) in-tree with hipcc.  No GPU needed (cross-compile).

    python -m flowmse_amd.build [--force]
"""
import hashlib
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libflowse_hip.so")
SOURCES = ["model_build.hip", "model_plan.hip", "model_api.hip", "conv_dispatch.hip", "conv_f43.hip", "conv_w2d.hip", "conv_halo.hip",
           "conv_flat.hip", "conv_1x1.hip", "conv_smallm.hip", "conv_reduce.hip", "conv16.hip", "conv16_pc.hip", "conv16_smallm.hip", "norm.hip", "fir.hip", "attention.hip", "misc.hip", "spec.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wall",
         "-Wno-unused-function"] + os.environ.get("FLOWSE_BUILD_FLAGS", "").split()   # extra flags for A-B builds (tools/build_variants.py)


# per-file extra flags.  conv16_pc.hip: its staging waves run beside an MFMA stream, where packed fp32 VALU instructions cost
# more than the scalar ones they replace -- keep the SLP vectoriser from re-packing the scalar code (explicit vector types
# in its output stage stay packed).
EXTRA_FLAGS = {"conv16_pc.hip": ["-fno-slp-vectorize"]}


def _digest():
    h = hashlib.sha256()
    files = [os.path.join(CSRC, n) for n in sorted(os.listdir(CSRC)) if not n.endswith(".o")]
    files.append(os.path.join(HERE, "..", "include", "flowse_hip.h"))
    for path in files:
        if os.path.isfile(path):
            h.update(os.path.basename(path).encode())
            with open(path, "rb") as f:
                h.update(f.read())
    h.update(" ".join(FLAGS).encode())
    h.update(repr(sorted(EXTRA_FLAGS.items())).encode())
    return h.hexdigest()


def build(force=False, verbose=True):
    stamp = LIB + ".stamp"
    dig = _digest()
    if not force and os.path.exists(LIB) and os.path.exists(stamp) and open(stamp).read() == dig:
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objs = []
    procs = []
    for src in SOURCES:
        obj = os.path.join(CSRC, src.replace(".hip", ".o"))
        objs.append(obj)
        cmd = [hipcc] + FLAGS + EXTRA_FLAGS.get(src, []) + ["-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{out.decode()}")
        if verbose and out.strip():
            print(out.decode())
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    with open(stamp, "w") as f:
        f.write(dig)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))

#!/bin/bash
# what-if probe builds of the 2-D Winograd kernel (tools/build_variants.py; results of probe builds are garbage, only
# their timing means something): kernel time of ONE shape per variant
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
IDX=${1:-0}
for v in base noout nostage nogn noxf nobar all3; do
  if [ "$v" != base ]; then export FLOWSE_LIB_PATH=flowmse_amd/variants/$v/libflowse_hip.so; else unset FLOWSE_LIB_PATH; fi
  rm -rf gpurun_out/prof_probe
  timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_probe -o p --output-format csv -- python tools/bench_wino.py --gn --only $IDX --form w2d --iters 6 > gpurun_out/probe_$v.log 2>&1
  python - "$v" <<'PY'
import csv, glob, sys
f = glob.glob("gpurun_out/prof_probe/**/*kernel_trace.csv", recursive=True)
d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in csv.DictReader(open(f[0])) if "w2d_kernel" in r["Kernel_Name"]]
print(f"{sys.argv[1]:10s} n={len(d)} min {min(d):8.1f} med {sorted(d)[len(d)//2]:8.1f} us")
PY
done

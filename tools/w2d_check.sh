#!/bin/bash
# 2-D Winograd kernel: parity tests, A-B micro-benchmark and a kernel trace of it (gpurun)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_ops.py -x -q -k "winograd" -s > gpurun_out/w2d_tests.log 2>&1; echo "tests rc=$?"; grep -E "winograd|passed|failed|Error|error" gpurun_out/w2d_tests.log | tail -30
rm -rf gpurun_out/prof_w2d
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_w2d -o w2d --output-format csv -- python tools/bench_wino.py --gn --iters 5 > gpurun_out/w2d_bench.log 2>&1; echo "bench rc=$?"
cat gpurun_out/w2d_bench.log | grep -v "^W\|rocprof" | tail -16
python - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/prof_w2d/**/*kernel_stats.csv", recursive=True)
if f:
    rows = list(csv.DictReader(open(f[0])))
    for r in rows[:14]:
        print(f"{r['Name'][:90]:90s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:9.1f} us  {r['Percentage']}%")
PY

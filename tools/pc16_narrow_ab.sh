#!/bin/bash
# A-B of conv3x3_pc16_kernel's 64-channel blocks (NJ = 1) on the bf16 storage mode: parity cases first, then the bench line
# with the policy off / on (and the hipGraph replay beside it).  Run through gpurun; outputs under gpurun_out/.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "16bit" > gpurun_out/nj_tests.log 2>&1; tail -3 gpurun_out/nj_tests.log
run() {  # tag, env...
    local tag=$1; shift
    env "$@" timeout 600 python bench.py --steps 10 --warmup 3 --precision bf16 --no-alt --no-cpu-baseline > gpurun_out/nj_$tag.json 2> gpurun_out/nj_$tag.err
    python - <<PY
import json
j=[json.loads(l) for l in open('gpurun_out/nj_$tag.json') if l.startswith('{')][-1]
print('$tag', round(j['value']), j['ms_per_step'], 'mfma_frac', j['roofline']['frac'], 'avg_launch_ms', j['roofline']['avg_launch_ms'], j.get('rel_l2_vs_oracle'))
PY
}
run wide FLOWSE_PC16_NARROW=0
run auto FLOWSE_X=1
run wide2 FLOWSE_PC16_NARROW=0
run auto2 FLOWSE_X=1
run graph FLOWSE_GRAPH=1

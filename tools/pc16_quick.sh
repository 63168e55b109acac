#!/bin/bash
# 16-bit parity cases, bf16 / fp16 bench lines (twice each) and the one-item-per-block timestamp accounting: the loop for a
# change to conv3x3_pc16_kernel.  python tools/build_variants.py meas -DFLOWSE_MEASURE first for the timestamps.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_blocks.py -x -q -m gpu -k "16" > gpurun_out/q_tests.log 2>&1; tail -3 gpurun_out/q_tests.log
run() {  # tag precision env...
    local tag=$1 prec=$2; shift 2
    env "$@" timeout 600 python bench.py --steps 10 --warmup 3 --precision $prec --no-alt --no-cpu-baseline > gpurun_out/q_$tag.json 2> gpurun_out/q_$tag.err
    python - <<PY
import json
j=[json.loads(l) for l in open('gpurun_out/q_$tag.json') if l.startswith('{')][-1]
print('$tag', round(j['value']), j['ms_per_step'], 'mfma_frac', round(j['roofline']['frac'],4), 'avg_launch_ms', round(j['roofline']['avg_launch_ms'],5), j.get('rel_l2_vs_oracle'))
PY
}
run bf16_a bf16 X=1
run fp16_a fp16 X=1
run bf16_b bf16 X=1
if [ -f flowmse_amd/variants/meas/libflowse_hip.so ]; then
  export FLOWSE_LIB_PATH=flowmse_amd/variants/meas/libflowse_hip.so
  for cfg in "8 256 0 256 32 32 1 1 1" "8 256 0 256 64 64 1 1 1" "8 128 0 128 128 128 1 1 1"; do
    echo "== $cfg"; timeout 300 python tools/pc16_ts.py $cfg 2>&1 | grep -v "amdgpu.ids\|xcd"
  done
fi

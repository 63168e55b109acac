# 16-bit conv checks after a change to conv3x3_pc16_kernel: op parity, block / model parity in the 16-bit modes, the
# per-role accounting (needs the meas variant, see pc16_ts.sh) and the bf16 bench line with the per-op table.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "16bit" 2>&1 | tail -3
timeout 900 python -m pytest tests/test_gpu_blocks.py tests/test_gpu_model.py -m gpu -x -q -k "16 or precision or bf16 or fp16" 2>&1 | tail -3
if [ -f flowmse_amd/variants/meas/libflowse_hip.so ]; then
  for cfg in "8 128 0 128 256 256 1 1 1" "8 256 0 128 256 256 1 0 1"; do
    FLOWSE_LIB_PATH=flowmse_amd/variants/meas/libflowse_hip.so timeout 300 python tools/pc16_ts.py $cfg 2>&1 | grep -v amdgpu.ids
  done
fi
timeout 600 python bench.py --steps 5 --warmup 2 --precision bf16 --no-alt --no-cpu-baseline --profile-all > gpurun_out/bench_bf16_pc.json 2> gpurun_out/bench_bf16_pc.err; echo "bench rc=$?"
python - <<'PY'
import json
j=[json.loads(l) for l in open('gpurun_out/bench_bf16_pc.json') if l.startswith('{')][-1]
print('bf16', round(j['value']), j['ms_per_step'], 'mfma_frac', j['roofline']['frac'], 'avg_launch_ms', j['roofline']['avg_launch_ms'])
PY
grep "^# conv" gpurun_out/bench_bf16_pc.err | head -14

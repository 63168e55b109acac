cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "16bit" 2>&1 | tail -2
timeout 600 python -m pytest tests/test_gpu_blocks.py tests/test_gpu_model.py -m gpu -x -q -k "16 or precision or bf16 or fp16" 2>&1 | tail -2
export FLOWSE_LIB_PATH=flowmse_amd/variants/meas/libflowse_hip.so
for cfg in "8 128 0 128 256 256 1 1 1" "8 256 0 128 256 256 1 0 1"; do
  timeout 300 python tools/pc16_ts.py $cfg 2>&1 | grep -v amdgpu.ids
done
unset FLOWSE_LIB_PATH
timeout 600 python bench.py --steps 5 --warmup 2 --precision bf16 --no-alt --no-cpu-baseline --profile-all > gpurun_out/bench_bf16_pc.json 2> gpurun_out/bench_bf16_pc.err; echo "pc rc=$?"
python - <<'PY'
import json
for f in ('gpurun_out/bench_bf16_pc.json',):
    try:
        j=[json.loads(l) for l in open(f) if l.startswith('{')][-1]
        print(f, round(j['value']), j['ms_per_step'], j['roofline']['frac'], j['roofline']['avg_launch_ms'])
    except Exception as e: print(f, 'ERR', e)
PY
grep "^# conv" gpurun_out/bench_bf16_pc.err | head -12

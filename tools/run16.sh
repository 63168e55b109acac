cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_blocks.py tests/test_gpu_ops.py -m gpu -x -q > gpurun_out/r2_t16a.log 2>&1; tail -25 gpurun_out/r2_t16a.log
timeout 1500 python -m pytest tests/test_gpu_model.py -m gpu -x -q -k "16bit or precision or golden or graph or shapes" > gpurun_out/r2_t16b.log 2>&1; tail -25 gpurun_out/r2_t16b.log
timeout 300 python bench.py --steps 3 --warmup 2 --precision bf16 --no-cpu-baseline --no-alt --profile-all > gpurun_out/r2_bf16s_b8.json 2> gpurun_out/r2_bf16s_b8.err; head -40 gpurun_out/r2_bf16s_b8.err
timeout 300 python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-alt --profile-all > gpurun_out/r2_fp32_b8.json 2> gpurun_out/r2_fp32_b8.err; grep -E "fir|level" gpurun_out/r2_fp32_b8.err | head -20
python - <<'PY'
import json
for f in ('gpurun_out/r2_bf16s_b8.json','gpurun_out/r2_fp32_b8.json'):
    try:
        j=[json.loads(l) for l in open(f) if l.startswith('{')][-1]; print(f, round(j['value']), j['ms_per_step'])
    except Exception as e: print(f,'ERR',e)
PY

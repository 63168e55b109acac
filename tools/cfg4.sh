# BASELINE config[4] (RK4 N=25, [32,1,256,1024], fp16 storage, 97 NFE per step) without a profiler + the 16-bit parity subset
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "16bit" 2>&1 | tail -2
timeout 900 python bench.py --steps 1 --warmup 1 --solver rk4 --nsolver 25 --batch 32 --frames 1024 --precision fp16 --no-cpu-baseline --no-alt > gpurun_out/cfg4.json 2> gpurun_out/cfg4.err; echo "rc=$?"
python - <<'PY'
import json
j=[json.loads(l) for l in open('gpurun_out/cfg4.json') if l.startswith('{')][-1]
print('config[4]', round(j['value']), j['ms_per_step'], 'mfma_frac', j['roofline']['frac'], 'avg_launch_ms', j['roofline']['avg_launch_ms'], j.get('parity_note'))
PY

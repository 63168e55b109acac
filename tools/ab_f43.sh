cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "winograd" 2>&1 | tail -2
timeout 900 python -m pytest tests/test_gpu_model.py -m gpu -x -q -k "full" 2>&1 | tail -2
for v in 0 1; do
  timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-alt --profile-all > gpurun_out/abf.json 2> gpurun_out/abf.err
  python - <<PY
import json
j=[json.loads(l) for l in open('gpurun_out/abf.json') if l.startswith('{')][-1]
print('fp32', round(j['value']), round(j['ms_per_step'],2), 'frac', round(j['roofline']['frac'],3))
PY
  grep "conv1_3x3_gn@256x256\|conv0_3x3_gn@256x256:256\|conv1_3x3_gn@32x32" gpurun_out/abf.err
done

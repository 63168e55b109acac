cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "winograd or conv" 2>&1 | tail -3
timeout 1500 python -m pytest tests/test_gpu_model.py -m gpu -x -q -k "full" 2>&1 | tail -3
for v in 0 1; do
  if [ $v = 1 ]; then export FLOWSE_F43_BN64=1; else unset FLOWSE_F43_BN64; fi
  timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-alt --profile-all > gpurun_out/abf_$v.json 2> gpurun_out/abf_$v.err
  python - <<PY
import json
j=[json.loads(l) for l in open('gpurun_out/abf_$v.json') if l.startswith('{')][-1]
print('bn64=$v fp32', round(j['value']), round(j['ms_per_step'],2), 'frac', round(j['roofline']['frac'],3), j['roofline']['avg_launch_ms'])
PY
  grep "conv1_3x3_gn@256x256\|conv0_3x3_gn@256x256:256\|conv1_3x3_gn@64x64\|conv1_3x3_gn@128x128:128\|conv0_3x3@256" gpurun_out/abf_$v.err
done

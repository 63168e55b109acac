cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_blocks.py tests/test_gpu_ops.py -m gpu -x -q -k "16" 2>&1 | tail -4
timeout 900 python -m pytest tests/test_gpu_model.py -m gpu -x -q -k "16bit or precision or bf16x3" 2>&1 | tail -4
for v in 0 1; do
  if [ $v = 1 ]; then export FLOWSE_HALO16_MT1=1; else unset FLOWSE_HALO16_MT1; fi
  timeout 300 python bench.py --steps 4 --warmup 2 --precision bf16 --no-cpu-baseline --no-alt --profile-all > gpurun_out/r2_ab16_$v.json 2> gpurun_out/r2_ab16_$v.err
  python - <<PY
import json
j=[json.loads(l) for l in open('gpurun_out/r2_ab16_$v.json') if l.startswith('{')][-1]
print('mt1=$v bf16', round(j['value']), round(j['ms_per_step'],2))
PY
  grep "conv1_3x3_gn@256x256\|conv0_3x3_gn@256x256:256\|conv1_3x3_gn@64x64\|conv1_3x3_gn@128x128:128" gpurun_out/r2_ab16_$v.err
done

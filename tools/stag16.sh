cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; mkdir -p gpurun_out
for v in 0 1 2 4 8; do
  FLOWSE_STAGGER16=$v timeout 300 python bench.py --steps 3 --warmup 2 --precision bf16 --no-cpu-baseline --no-alt --profile-all > gpurun_out/stag_$v.json 2> gpurun_out/stag_$v.err
  python - <<PY
import json
j=[json.loads(l) for l in open('gpurun_out/stag_$v.json') if l.startswith('{')][-1]
print('stagger=$v bf16', round(j['value']), round(j['ms_per_step'],2))
PY
  echo "   $(grep 'conv1_3x3_gn@256x256:128>128' gpurun_out/stag_$v.err | awk '{print $4, $5}') | $(grep 'conv0_3x3_gn@256x256:256>128' gpurun_out/stag_$v.err | awk '{print $4,$5}') | $(grep 'conv1_3x3_gn@128x128:128>128' gpurun_out/stag_$v.err | awk '{print $4,$5}')"
done

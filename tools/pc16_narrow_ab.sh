cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
for v in auto 1 0 auto 1; do
  if [ $v = auto ]; then unset FLOWSE_PC16_NARROW; else export FLOWSE_PC16_NARROW=$v; fi
  timeout 600 python bench.py --steps 10 --warmup 3 --precision bf16 --no-alt --no-cpu-baseline > gpurun_out/n_$v.json 2>/dev/null
  python - <<PY
import json
j=[json.loads(l) for l in open('gpurun_out/n_$v.json') if l.startswith('{')][-1]
print('narrow=$v', round(j['value']), j['ms_per_step'])
PY
done

cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_model.py tests/test_gpu_blocks.py -m gpu -x -q 2>&1 | tail -5
for b in 1 8; do
 for tp in 0 1; do
  if [ $tp = 1 ]; then export FLOWSE_SPLITK_TWO_PASS=1; else unset FLOWSE_SPLITK_TWO_PASS; fi
  timeout 300 python bench.py --batch $b --steps 6 --warmup 3 --no-cpu-baseline --no-alt > gpurun_out/sk.json 2>gpurun_out/sk.err || tail -3 gpurun_out/sk.err
  python - <<PY
import json
j=[json.loads(l) for l in open('gpurun_out/sk.json') if l.startswith('{')][-1]
print('two_pass=$tp B=$b', round(j['value']), round(j['ms_per_step'],2), j['launches_per_nfe'])
PY
 done
done

cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; mkdir -p gpurun_out
export FLOWSE_BENCH_NO_CHECK=1
for skip in none splitk_reduce gn_norm gn_finalize "splitk_reduce,gn_norm,gn_finalize"; do
  for b in 1 8; do
    if [ "$skip" = none ]; then unset FLOWSE_SKIP_OPS; else export FLOWSE_SKIP_OPS=$skip; fi
    timeout 300 python bench.py --batch $b --steps 6 --warmup 3 --no-cpu-baseline --no-alt > gpurun_out/wi.json 2>gpurun_out/wi.err
    python - <<PY
import json
j=[json.loads(l) for l in open('gpurun_out/wi.json') if l.startswith('{')][-1]
print('skip=$skip B=$b', round(j['value']), round(j['ms_per_step'],2))
PY
  done
done

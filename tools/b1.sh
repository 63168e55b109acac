cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 300 python bench.py --batch 1 --steps 10 --warmup 3 --no-cpu-baseline --no-alt --profile-all > gpurun_out/r2_b1.json 2> gpurun_out/r2_b1.err
python - <<'PY'
import json
j=[json.loads(l) for l in open('gpurun_out/r2_b1.json') if l.startswith('{')][-1]
print('fp32 B=1', round(j['value']), j['ms_per_step'])
PY
grep -v "^/opt" gpurun_out/r2_b1.err | head -70

cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; mkdir -p gpurun_out
for v in 1024 512 256 1024 512; do
  export FLOWSE_F43_WIDE_MIN=$v
  timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-alt > gpurun_out/abf_$v.json 2> gpurun_out/abf_$v.err
  python - <<PY
import json
j=[json.loads(l) for l in open('gpurun_out/abf_$v.json') if l.startswith('{')][-1]
print('wide_min=$v B=8', round(j['value']), round(j['ms_per_step'],2), 'frac', round(j['roofline']['frac'],3))
PY
done
for v in 1024 512 256 128; do
  export FLOWSE_F43_WIDE_MIN=$v
  timeout 300 python bench.py --batch 1 --steps 8 --warmup 3 --no-cpu-baseline --no-alt > gpurun_out/abf1_$v.json 2> gpurun_out/abf1_$v.err
  python - <<PY
import json
j=[json.loads(l) for l in open('gpurun_out/abf1_$v.json') if l.startswith('{')][-1]
print('wide_min=$v B=1', round(j['value']), round(j['ms_per_step'],2))
PY
done

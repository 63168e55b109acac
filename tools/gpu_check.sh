#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -6 gpurun_out/r2_pytest3.log
timeout 900 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --profile-all > gpurun_out/bench.json 2> gpurun_out/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
j=[json.loads(l) for l in open('gpurun_out/bench.json') if l.startswith('{')][-1]
print('fp32', round(j['value']), j['ms_per_step'], 'frac', j['roofline']['frac'], 'avg_launch_ms', j['roofline']['avg_launch_ms'])
for k,v in j.get('alt_precision',{}).items():
    print(k, round(v['value']), round(v['ms_per_step'],2), v['rel_l2_vs_fp32_mode'], {a:(round(b,3) if isinstance(b,float) else b) for a,b in v.get('roofline',{}).items() if a in ('bound','mfma_frac','hbm_frac','avg_launch_ms')})
PY
head -30 gpurun_out/bench.err

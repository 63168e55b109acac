#!/bin/bash
# full `pytest -m gpu` + default bench with the per-op table; outputs under gpurun_out/
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -6 gpurun_out/pytest_gpu.log
timeout 900 python bench.py --steps 5 --warmup 2 --profile-all > gpurun_out/bench.json 2> gpurun_out/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
j=[json.loads(l) for l in open('gpurun_out/bench.json') if l.startswith('{')][-1]
print('fp32', round(j['value']), j['ms_per_step'], 'frac', j['roofline']['frac'], 'avg_launch_ms', j['roofline']['avg_launch_ms'], 'whole', j['roofline'].get('whole_path_issued_frac'), 'launches/nfe', j.get('launches_per_nfe'))
for k,v in j.get('alt_precision',{}).items():
    print(k, round(v['value']), round(v['ms_per_step'],2), v['rel_l2_vs_fp32_mode'], v.get('rel_l2_vs_oracle'), {a:(round(b,3) if isinstance(b,float) else b) for a,b in v.get('roofline',{}).items() if a in ('bound','mfma_frac','hbm_frac','avg_launch_ms')})
print('alt_shapes', j.get('alt_shapes'))
w=j.get('alt_workloads',{}).get('vbdmd_one_gpu_share',{})
print('vbdmd share', w.get('value'), w.get('vs_headline_rate'), w.get('plan'))
print('rel_l2_vs_oracle', j.get('rel_l2_vs_oracle'))
print('cpu', j.get('cpu_baseline',{}).get('value'), j.get('gpu_vs_cpu'))
PY
head -40 gpurun_out/bench.err

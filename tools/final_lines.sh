# the bench lines committed under profiles/ for a round: the driver's command, the config[3] workload on one GPU
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1200 python bench.py --steps 20 --warmup 5 > gpurun_out/bench_line.json 2> gpurun_out/bench_line.err; echo "bench rc=$?"
timeout 900 python bench.py --workload vbdmd --steps 2 --warmup 2 > gpurun_out/vbdmd_line.json 2> gpurun_out/vbdmd_line.err; echo "vbdmd rc=$?"
python - <<'PY'
import json
j=[json.loads(l) for l in open('gpurun_out/bench_line.json') if l.startswith('{')][-1]
print('fp32', round(j['value']), j['ms_per_step'], 'frac', j['roofline']['frac'], 'whole', j['roofline'].get('whole_path_issued_frac'))
for k,v in j.get('alt_precision',{}).items(): print(k, round(v['value']), v.get('rel_l2_vs_oracle'), v.get('roofline',{}).get('mfma_frac'))
print(j.get('alt_shapes')); w=j.get('alt_workloads',{}).get('vbdmd_one_gpu_share',{}); print('share', w.get('value'), w.get('vs_headline_rate'))
v=[json.loads(l) for l in open('gpurun_out/vbdmd_line.json') if l.startswith('{')][-1]
print('vbdmd', round(v['value']), v['ms_per_step'])
PY

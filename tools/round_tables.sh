#!/bin/bash
# per-op tables committed under profiles/ (bench.py --profile-all writes them to stderr): fp32 and bf16 at the headline shape,
# fp32 for a single utterance; and BASELINE config[4]'s bench line
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python bench.py --steps 5 --warmup 2 --no-alt --no-cpu-baseline --profile-all > gpurun_out/tab_fp32.json 2> gpurun_out/tab_fp32.err
timeout 600 python bench.py --steps 5 --warmup 2 --precision bf16 --no-alt --no-cpu-baseline --profile-all > gpurun_out/tab_bf16.json 2> gpurun_out/tab_bf16.err
timeout 600 python bench.py --batch 1 --steps 5 --warmup 2 --no-alt --no-cpu-baseline --profile-all > gpurun_out/tab_fp32_b1.json 2> gpurun_out/tab_fp32_b1.err
timeout 900 python bench.py --steps 2 --warmup 1 --solver rk4 --nsolver 25 --batch 32 --frames 1024 --precision fp16 --no-cpu-baseline --no-alt > gpurun_out/cfg4_line.json 2> gpurun_out/cfg4_line.err
for t in fp32 bf16 fp32_b1; do grep "^# level" gpurun_out/tab_$t.err | head -8; done
python - <<'PY'
import json
j=[json.loads(l) for l in open('gpurun_out/cfg4_line.json') if l.startswith('{')][-1]
print('cfg4', round(j['value']), j['ms_per_step'], j['roofline'].get('frac'), j['roofline'].get('avg_launch_ms'))
PY

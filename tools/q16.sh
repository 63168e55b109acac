#!/bin/bash
# 16-bit loop: every 16-bit parity test (ops, blocks, network), then bf16 / fp16 lines and the launch count
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1200 python -m pytest tests -x -q -m gpu -k "16 or bf16 or fp16 or combine" > gpurun_out/q16_tests.log 2>&1; tail -2 gpurun_out/q16_tests.log
for p in bf16 fp16 bf16; do
timeout 600 python bench.py --steps 10 --warmup 3 --precision $p --no-alt --no-cpu-baseline > gpurun_out/q16_$p.json 2>/dev/null
python - <<PY
import json
j=[json.loads(l) for l in open('gpurun_out/q16_$p.json') if l.startswith('{')][-1]
print('$p', round(j['value']), j['ms_per_step'], j.get('launches_per_nfe'), j.get('rel_l2_vs_oracle'))
PY
done

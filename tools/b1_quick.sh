#!/bin/bash
# single-utterance loop: GroupNorm / block parity tests, then the [1,1,256,256] bench line with its gn_norm per-op lines
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests -x -q -m gpu -k "group_norm or resblock or attnblock or B1 or single" > gpurun_out/b1_tests.log 2>&1; tail -2 gpurun_out/b1_tests.log
for p in fp32 bf16; do
timeout 600 python bench.py --batch 1 --steps 10 --warmup 3 --precision $p --no-alt --no-cpu-baseline > gpurun_out/b1q_$p.json 2>/dev/null
python - <<PY
import json
j=[json.loads(l) for l in open('gpurun_out/b1q_$p.json') if l.startswith('{')][-1]
print('$p B=1', round(j['value']), j['ms_per_step'], j.get('launches_per_nfe'))
PY
done
timeout 600 python bench.py --batch 1 --steps 3 --warmup 2 --no-alt --no-cpu-baseline --profile-all 2>&1 | grep "gn_norm@\|gn_finalize@" | head -12

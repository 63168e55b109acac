#!/bin/bash
# block / network parity tests, then fp32 lines at batch 8 and batch 1 with chosen per-op lines: PAT="combine|gn_norm" bash tools/quick_all.sh
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_blocks.py tests/test_gpu_model.py -x -q -m gpu -k "not switches and not gpus" > gpurun_out/qa_tests.log 2>&1; tail -2 gpurun_out/qa_tests.log
for b in 8 1; do
  timeout 600 python bench.py --batch $b --steps 10 --warmup 3 --no-alt --no-cpu-baseline > gpurun_out/qa_$b.json 2>/dev/null
  python - <<PY
import json
j=[json.loads(l) for l in open('gpurun_out/qa_$b.json') if l.startswith('{')][-1]
print('fp32 B=$b', round(j['value']), j['ms_per_step'], j.get('launches_per_nfe'))
PY
done
timeout 600 python bench.py --steps 3 --warmup 2 --no-alt --no-cpu-baseline --profile-all 2>&1 | grep -E "${PAT:-combine}" | head -16

#!/bin/bash
# attention kernels: parity tests + per-op time (fp32 batch 8 / batch 1, bf16) from bench.py --profile-all
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests -x -q -m gpu -k "attention or attn" > gpurun_out/at_tests.log 2>&1; tail -2 gpurun_out/at_tests.log
for cfg in "fp32 8" "fp32 1" "bf16 8"; do
  set -- $cfg
  timeout 600 python bench.py --steps 3 --warmup 2 --precision $1 --batch $2 --no-alt --no-cpu-baseline --profile-all > gpurun_out/at_$1_$2.json 2> gpurun_out/at_$1_$2.err
  echo "== $cfg"; grep "attention@\|attn_" gpurun_out/at_$1_$2.err
  python - <<PY
import json
j=[json.loads(l) for l in open('gpurun_out/at_$1_$2.json') if l.startswith('{')][-1]
print('value', round(j['value']), j['ms_per_step'])
PY
done

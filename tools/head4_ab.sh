#!/bin/bash
# 16-bit progressive-output heads: parity cases, bf16 / fp16 A-B against the fp32-widening form (FLOWSE_HEAD4_FP32=1), per-op lines
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1200 python -m pytest tests -x -q -m gpu -k "16 or bf16 or fp16" > gpurun_out/h4_tests.log 2>&1; tail -3 gpurun_out/h4_tests.log; grep "head4" gpurun_out/h4_tests.log | head
run() {
  env $3 timeout 600 python bench.py --steps 10 --warmup 3 --precision $2 --no-alt --no-cpu-baseline > gpurun_out/h4_$1.json 2>/dev/null
  python - <<PY
import json
j=[json.loads(l) for l in open('gpurun_out/h4_$1.json') if l.startswith('{')][-1]
print('$1', round(j['value']), j['ms_per_step'], j.get('rel_l2_vs_oracle'))
PY
}
run old_bf16 bf16 FLOWSE_HEAD4_FP32=1
run new_bf16 bf16 X=1
run old_bf16b bf16 FLOWSE_HEAD4_FP32=1
run new_bf16b bf16 X=1
run new_fp16 fp16 X=1
for v in 1 0; do
  echo "== fp32-widening=$v"; if [ $v = 1 ]; then export FLOWSE_HEAD4_FP32=1; else unset FLOWSE_HEAD4_FP32; fi
  timeout 600 python bench.py --steps 3 --warmup 2 --precision bf16 --no-alt --no-cpu-baseline --profile-all 2>&1 | grep "pyramid_conv@" | head -4
done

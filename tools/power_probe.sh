#!/bin/bash
# Is the 16-bit conv power-capped?  rocm-smi power / sclk samples (every 0.2 s) while bench.py runs one precision mode.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
rocm-smi --showpower --showclocks --showmaxpower 2>&1 | grep -v "^=\|^$" | head -20
for prec in ${PRECS:-bf16 fp32}; do
  ( while true; do rocm-smi --showpower --showclocks --json 2>/dev/null | tr -d '\n'; echo; sleep 0.2; done ) > gpurun_out/power_$prec.jsonl &
  SPID=$!
  timeout 600 python bench.py --steps 60 --warmup 5 --precision $prec --no-alt --no-cpu-baseline > gpurun_out/power_bench_$prec.json 2>/dev/null
  kill $SPID
  python - <<PY
import json
rows=[]
for l in open('gpurun_out/power_$prec.jsonl'):
    try: j=json.loads(l)
    except Exception: continue
    c=j.get('card0',{})
    p=[v for k,v in c.items() if 'ower' in k]
    s=[v for k,v in c.items() if 'sclk' in k]
    rows.append((p,s))
print('$prec', len(rows), 'samples')
for r in rows[::max(1,len(rows)//25)]: print('  ', r)
b=[json.loads(l) for l in open('gpurun_out/power_bench_$prec.json') if l.startswith('{')][-1]
print('  value', round(b['value']), 'frac', b['roofline']['frac'])
PY
done

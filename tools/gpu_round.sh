#!/bin/bash
# one GPU-box session: tests, bench, B=1 profile (outputs under gpurun_out/)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r2_pytest.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/r2_pytest.log
tail -5 gpurun_out/r2_pytest.log
timeout 600 python bench.py --steps 5 --warmup 2 > gpurun_out/r2_bench_b8.json 2> gpurun_out/r2_bench_b8.err; echo "bench rc=$?"
timeout 300 env FLOWSE_NO_GRAPH=1 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-alt > gpurun_out/r2_bench_b8_nograph.json 2>&1
timeout 300 python bench.py --steps 10 --warmup 3 --batch 1 --no-cpu-baseline --no-alt --profile-all > gpurun_out/r2_bench_b1.json 2> gpurun_out/r2_bench_b1.err
timeout 300 env FLOWSE_NO_GRAPH=1 python bench.py --steps 10 --warmup 3 --batch 1 --no-cpu-baseline --no-alt > gpurun_out/r2_bench_b1_nograph.json 2>&1
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r2_bench_*.json')):
    try:
        j=[json.loads(l) for l in open(f) if l.startswith('{')][-1]
        print(f, round(j['value']), 'frames/s', round(j['ms_per_step'],2),'ms', 'launches/nfe', j.get('launches_per_nfe'), 'frac', j.get('roofline',{}).get('frac'))
    except Exception as e:
        print(f, 'ERR', e)
PY

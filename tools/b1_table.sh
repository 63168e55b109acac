# per-op table of the single-utterance shape ([1,1,256,256]) in fp32 and bf16: where the small-batch time goes
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
for p in fp32 bf16; do
  timeout 600 python bench.py --batch 1 --steps 5 --warmup 2 --precision $p --no-alt --no-cpu-baseline --profile-all > gpurun_out/b1_$p.json 2> gpurun_out/b1_$p.err; echo "$p rc=$?"
  python - <<PY
import json
j=[json.loads(l) for l in open('gpurun_out/b1_$p.json') if l.startswith('{')][-1]
print('$p', round(j['value']), j['ms_per_step'], j.get('launches_per_nfe'))
PY
done

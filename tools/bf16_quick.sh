# bf16 / fp16 bench lines with the per-op table (no parity tests): a quick A-B after a 16-bit kernel or policy change
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
for p in bf16 fp16; do
timeout 600 python bench.py --steps 5 --warmup 2 --precision $p --no-alt --no-cpu-baseline --profile-all > gpurun_out/q_$p.json 2> gpurun_out/q_$p.err; echo "$p rc=$?"
python - <<PY
import json
j=[json.loads(l) for l in open('gpurun_out/q_$p.json') if l.startswith('{')][-1]
print('$p', round(j['value']), j['ms_per_step'], 'mfma_frac', j['roofline']['frac'], 'avg_launch_ms', j['roofline']['avg_launch_ms'], j.get('rel_l2_vs_oracle'))
PY
done
grep "^# level\|32x32\|64x64" gpurun_out/q_bf16.err | head -40

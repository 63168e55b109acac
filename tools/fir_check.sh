# FIR kernels after a change: parity (ops + resampling ResnetBlocks + whole net) and their lines of the per-op tables
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_blocks.py -m gpu -x -q -k "fir or upfirdn or resblock" 2>&1 | tail -2
timeout 900 python -m pytest tests/test_gpu_model.py -m gpu -x -q -k "golden or oracle_B8 or precision" 2>&1 | tail -2
for p in fp32 bf16; do
timeout 600 python bench.py --steps 5 --warmup 2 --precision $p --no-alt --no-cpu-baseline --profile-all > gpurun_out/f_$p.json 2> gpurun_out/f_$p.err; echo "$p rc=$?"
python - <<PY
import json
j=[json.loads(l) for l in open('gpurun_out/f_$p.json') if l.startswith('{')][-1]
print('$p', round(j['value']), j['ms_per_step'])
PY
grep "fir_" gpurun_out/f_$p.err | head -8
done

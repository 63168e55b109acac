cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; mkdir -p gpurun_out
FLOWSE_NO_GRAPH=1 timeout 300 python bench.py --steps 1 --warmup 1 --precision bf16 --no-cpu-baseline --no-alt 2>&1 | tail -3 | cut -c1-300

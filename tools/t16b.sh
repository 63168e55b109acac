cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_blocks.py -m gpu -x -q -k "16x16_tile" -s 2>&1 | grep -v "^$" | tail -8

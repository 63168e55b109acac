cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; mkdir -p gpurun_out
FLOWSE_HALO16_MT1=1 timeout 600 python tools/ts16.py > gpurun_out/ts16_mt1.log 2>&1; grep -v "^{" gpurun_out/ts16_mt1.log | tail -32
timeout 600 python tools/ts16.py > gpurun_out/ts16_mt2.log 2>&1; grep -v "^{" gpurun_out/ts16_mt2.log | tail -32

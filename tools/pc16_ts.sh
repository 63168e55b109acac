cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
export FLOWSE_LIB_PATH=flowmse_amd/variants/meas/libflowse_hip.so
for cfg in "8 128 0 128 256 256 1 1 1" "8 256 0 128 256 256 1 0 1"; do
timeout 300 python tools/pc16_ts.py $cfg 2>&1 | grep -v "amdgpu.ids\|xcd"
done

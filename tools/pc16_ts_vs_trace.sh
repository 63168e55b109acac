cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
export FLOWSE_LIB_PATH=flowmse_amd/variants/meas/libflowse_hip.so
for cfg in "8 128 0 128 128 128 1 1 1" "8 256 0 256 64 64 1 1 1" "8 128 0 128 256 256 1 1 1"; do
rm -rf gpurun_out/prof_x
timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_x -o x --output-format csv -- python tools/pc16_ts.py $cfg 2>&1 | grep -v "amdgpu.ids\|xcd" | grep "entry ->\|blocks\|us per call\|100 MHz"
f=$(find gpurun_out/prof_x -name "*kernel_stats.csv" | head -1); grep "pc16_kernel" $f | cut -c1-140
done

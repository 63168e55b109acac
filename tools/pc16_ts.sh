# Per-role cycle accounting of conv3x3_pc16_kernel on the GPU box:
#   python tools/build_variants.py meas -DFLOWSE_MEASURE      (here, before gpurun: the variant travels with the snapshot)
#   gpurun --timeout 900 -- 'bash tools/pc16_ts.sh'
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
export FLOWSE_LIB_PATH=flowmse_amd/variants/meas/libflowse_hip.so
[ -f "$FLOWSE_LIB_PATH" ] || { echo "build the measurement variant first (see header)"; exit 1; }
# B C1 C2 Cout H W gn res silu
for cfg in "8 128 0 128 256 256 1 1 1" "8 256 0 128 256 256 1 0 1" "8 128 0 128 256 256 0 0 0"; do
  timeout 300 python tools/pc16_ts.py $cfg 2>&1 | grep -v amdgpu.ids
done

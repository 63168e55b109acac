# Per-role cycle accounting of conv3x3_pc16_kernel on the one-item-per-block launches (32 x 32 and 64 x 64 at batch 8), 64- vs
# 128-channel blocks.  python tools/build_variants.py meas -DFLOWSE_MEASURE first; gpurun -- 'bash tools/pc16_ts_small.sh'
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
export FLOWSE_LIB_PATH=flowmse_amd/variants/meas/libflowse_hip.so
for cfg in "8 256 0 256 32 32 1 1 1" "8 256 0 256 64 64 1 1 1"; do
  echo "== narrow/auto $cfg"; timeout 300 python tools/pc16_ts.py $cfg 2>&1 | grep -v amdgpu.ids
  echo "== wide $cfg"; FLOWSE_PC16_NARROW=0 timeout 300 python tools/pc16_ts.py $cfg 2>&1 | grep -v amdgpu.ids
done

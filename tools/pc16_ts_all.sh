#!/bin/bash
# per-role s_memtime accounting of conv3x3_pc16_kernel (variants/meas = -DFLOWSE_MEASURE build) on the shapes of DESIGN section 5
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
export FLOWSE_LIB_PATH=flowmse_amd/variants/meas/libflowse_hip.so
for cfg in "8 128 0 128 256 256 1 1 1" "8 128 0 128 256 256 1 0 1" "8 256 0 128 256 256 1 0 1" "8 128 0 128 128 128 1 1 1" "8 256 0 256 64 64 1 1 1" "8 256 0 256 32 32 1 1 1"; do
  echo "== $cfg"; timeout 300 python tools/pc16_ts.py $cfg 2>&1 | grep -v "amdgpu.ids\|xcd"
done

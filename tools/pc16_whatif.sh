#!/bin/bash
# what-if builds of conv3x3_pc16_kernel (timing only): per-role accounting on two shapes
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
for v in ${VARIANTS:-meas m_notrans m_noburst m_prio}; do
  export FLOWSE_LIB_PATH=flowmse_amd/variants/$v/libflowse_hip.so
  for cfg in "8 128 0 128 256 256 1 0 1" "8 256 0 256 32 32 1 0 1"; do
    echo "== $v: $cfg"; timeout 300 python tools/pc16_ts.py $cfg 2>&1 | grep -v "amdgpu.ids\|xcd"
  done
done

#!/bin/bash
# parity subset under every library switch that selects a different kernel path
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
for e in "" FLOWSE_F43_BN64=1 FLOWSE_WINOGRAD=f23 FLOWSE_NO_WINOGRAD=1 FLOWSE_NO_GRAPH=1 FLOWSE_SPLITK_IN_LAUNCH=1 FLOWSE_HALO16_MT1=1 FLOWSE_HALO16_PER_TAP=1 FLOWSE_NO_HALO16=1; do
  r=$(env $e timeout 900 python -m pytest tests/test_gpu_model.py -m gpu -x -q -k "full or tiny or precision or 16bit" 2>&1 | tail -1)
  echo "[$e] $r"
done

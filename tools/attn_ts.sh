cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
FLOWSE_LIB_PATH=flowmse_amd/variants/attts/libflowse_hip.so timeout 300 python bench.py --steps 1 --warmup 1 --no-alt --no-cpu-baseline 2>&1 | grep "^ATT" | sort | uniq -c | sort -rn | head -12
python tools/ab.py --bench-args "--steps 5 --warmup 2 --no-alt --no-cpu-baseline --profile-all" base h4acc2 2>&1 | tail -3
for v in base h4acc2; do
  if [ $v = base ]; then unset FLOWSE_LIB_PATH; else export FLOWSE_LIB_PATH=flowmse_amd/variants/$v/libflowse_hip.so; fi
  for p in fp32 bf16; do timeout 300 python bench.py --steps 3 --warmup 2 --precision $p --no-alt --no-cpu-baseline --profile-all 2>&1 | grep "pyramid_conv@256\|pyramid_conv@128"; done
done

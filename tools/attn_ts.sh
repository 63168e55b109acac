# phase stamps of attention_kernel (a throw-away variant build with s_memrealtime + printf: see the round-5 log in DESIGN.md)
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
FLOWSE_LIB_PATH=flowmse_amd/variants/attts/libflowse_hip.so timeout 300 python bench.py --steps 1 --warmup 1 --no-alt --no-cpu-baseline 2>&1 | grep "^ATT" | sort | uniq -c | sort -rn | head -8

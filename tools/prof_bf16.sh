cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
rm -rf gpurun_out/prof_bf16
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_bf16 -o x --output-format csv -- python bench.py --steps 2 --warmup 1 --precision bf16 --no-cpu-baseline --no-alt > gpurun_out/prof_bf16.log 2>&1
f=$(find gpurun_out/prof_bf16 -name "*kernel_stats.csv" | head -1); head -30 $f | cut -c1-150

#!/bin/bash
# rocprofv3 evidence for one round (run on the GPU box via gpurun; outputs under gpurun_out/prof_*; summarise with
# tools/summarize_profiles.py <tag>).  Counters in their own passes, only --kernel-trace beside them.
TAG=${1:-r04}
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
CMD="python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-alt"
CMD1="python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-alt"
rm -rf gpurun_out/prof_stats gpurun_out/prof_fetch gpurun_out/prof_write gpurun_out/prof_sq gpurun_out/prof_bf16 gpurun_out/prof_cfg5 gpurun_out/prof_sq16
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_stats -o $TAG --output-format csv -- $CMD > gpurun_out/prof_stats.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/prof_fetch -o $TAG --output-format csv -- $CMD1 > gpurun_out/prof_fetch.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/prof_write -o $TAG --output-format csv -- $CMD1 > gpurun_out/prof_write.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE -d gpurun_out/prof_sq -o $TAG --output-format csv -- $CMD1 > gpurun_out/prof_sq.log 2>&1
# BASELINE config[2]-shaped: bf16 storage, N=5 Euler, [8,1,256,256]
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_bf16 -o $TAG --output-format csv -- python bench.py --steps 2 --warmup 1 --precision bf16 --no-cpu-baseline --no-alt > gpurun_out/prof_bf16.log 2>&1
# SQ counters of the 16-bit producer / consumer conv (own pass)
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE -d gpurun_out/prof_sq16 -o $TAG --output-format csv -- python bench.py --steps 1 --warmup 1 --precision bf16 --no-cpu-baseline --no-alt > gpurun_out/prof_sq16.log 2>&1
# BASELINE config[4]: N=25 RK4, batch 32, T=1024, fp16 storage (97 NFE per step)
timeout 900 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_cfg5 -o $TAG --output-format csv -- python bench.py --steps 1 --warmup 1 --solver rk4 --nsolver 25 --batch 32 --frames 1024 --precision fp16 --no-cpu-baseline --no-alt > gpurun_out/prof_cfg5.log 2>&1
tail -2 gpurun_out/prof_cfg5.log; tail -1 gpurun_out/prof_bf16.log | cut -c1-300
find gpurun_out/prof_* -name "*.csv" | head -20

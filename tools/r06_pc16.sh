#!/bin/bash
# Round-6 loop for conv3x3_pc16_kernel: 16-bit parity subset, then A-B-A-B of the shipped build against the round-5 library
# kept under flowmse_amd/variants/r05 (bf16 and fp16 at [8,1,256,256]).
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_blocks.py -x -q -m gpu -k "16" > gpurun_out/q_tests.log 2>&1; tail -5 gpurun_out/q_tests.log
[ -n "$SKIP_MODEL" ] || { timeout 600 python -m pytest tests/test_gpu_model.py -x -q -m gpu -k "16bit" > gpurun_out/q_tests_model.log 2>&1; tail -3 gpurun_out/q_tests_model.log; }
python tools/ab.py --bench-args "--steps 10 --warmup 3 --precision bf16 --no-alt --no-cpu-baseline" ${VARIANTS:-r05 base r05 base} 2>&1 | tee gpurun_out/q_ab_bf16.log
python tools/ab.py --bench-args "--steps 10 --warmup 3 --precision fp16 --no-alt --no-cpu-baseline" ${VARIANTS:-r05 base} 2>&1 | tee gpurun_out/q_ab_fp16.log

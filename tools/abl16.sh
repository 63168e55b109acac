cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; mkdir -p gpurun_out
for v in 0 8 15 31 12 9; do
  FLOWSE_ABL16=$v timeout 300 python bench.py --steps 2 --warmup 2 --precision bf16 --no-cpu-baseline --no-alt --profile-all > gpurun_out/abl_$v.json 2> gpurun_out/abl_$v.err
  echo "ABL=$v $(grep 'conv1_3x3_gn@256x256:128>128' gpurun_out/abl_$v.err | awk '{print $4, $5}') | $(grep 'conv0_3x3_gn@256x256:256>128' gpurun_out/abl_$v.err | awk '{print $4,$5}') | $(tail -2 gpurun_out/abl_$v.err | head -1 | cut -c1-80)"
done

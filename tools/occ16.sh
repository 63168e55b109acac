cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; mkdir -p gpurun_out
export FLOWSE_HALO16_MT1=1
for v in 0 60000 100000; do
  FLOWSE_HALO16_LDS=$v timeout 300 python bench.py --steps 3 --warmup 2 --precision bf16 --no-cpu-baseline --no-alt --profile-all > gpurun_out/occ_$v.json 2> gpurun_out/occ_$v.err
  echo "LDS=$v: $(grep 'conv1_3x3_gn@256x256:128>128' gpurun_out/occ_$v.err | awk '{print $4, $5}') | $(grep 'conv0_3x3_gn@256x256:256>128' gpurun_out/occ_$v.err | awk '{print $4,$5}') | $(grep 'conv1_3x3_gn@128x128:128>128' gpurun_out/occ_$v.err | awk '{print $4,$5}')"
done

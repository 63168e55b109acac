#!/bin/bash
# SQ counter passes over ONE shape of tools/bench_wino.py (default: index 0 = [8,256,256,128->128]) for both Winograd
# kernels (one rocprofv3 run per counter group, only --kernel-trace beside --pmc).  Summary: gpurun_out/pmc_w2d/summary.txt
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
IDX=${1:-0}
OUT=gpurun_out/pmc_w2d
rm -rf $OUT; mkdir -p $OUT
run() {  # name, counters...
  name=$1; shift
  timeout 300 rocprofv3 --kernel-trace --pmc "$@" -d $OUT/$name -o p --output-format csv -- python tools/bench_wino.py --only $IDX --iters 3 > $OUT/$name.log 2>&1 || echo "pass $name failed"
}
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES
run sq2 SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVES
run sq3 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM
run sq4 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE
python - <<'PY'
import csv, glob, collections, os
out = open('gpurun_out/pmc_w2d/summary.txt', 'w')
dur = collections.defaultdict(list)
for f in glob.glob('gpurun_out/pmc_w2d/sq1/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        if 'f43_kernel' in k or 'w2d_kernel' in k:
            dur[k[:44]].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
for k, v in dur.items():
    line = f"duration {k} n={len(v)} min={min(v):.1f} us med={sorted(v)[len(v)//2]:.1f} us"
    print(line); out.write(line + "\n")
for d in sorted(glob.glob('gpurun_out/pmc_w2d/*/')):
    for f in glob.glob(d + '**/*counter_collection.csv', recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            k = r['Kernel_Name']
            if 'f43_kernel' not in k and 'w2d_kernel' not in k: continue
            acc[k[:44]][r['Counter_Name']].append(float(r['Counter_Value']))
        for key, cs in sorted(acc.items()):
            for c, v in sorted(cs.items()):
                line = f"{os.path.basename(os.path.dirname(d))} {key} {c} n={len(v)} mean={sum(v)/len(v):.6e}"
                print(line); out.write(line + "\n")
PY

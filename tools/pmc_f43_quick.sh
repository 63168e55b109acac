#!/bin/bash
# two quick PMC passes over the dominant kernel alone (MFMA busy, instruction mix)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
OUT=gpurun_out/pmc_f43q
rm -rf $OUT; mkdir -p $OUT
run() { name=$1; shift
  timeout 300 rocprofv3 --kernel-trace --pmc "$@" -d $OUT/$name -o p --output-format csv -- python tools/f43_probe.py 3 > $OUT/$name.log 2>&1 || echo "pass $name failed"; }
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE
run sq2 SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVES
python - <<'PY'
import csv, glob, collections, os
for d in sorted(glob.glob('gpurun_out/pmc_f43q/*/')):
    for f in glob.glob(d + '**/*counter_collection.csv', recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            k = r['Kernel_Name']
            if 'f43_kernel' not in k: continue
            acc[(k[:48], r.get('Grid_Size', '?'))][r['Counter_Name']].append(float(r['Counter_Value']))
        for key, cs in sorted(acc.items()):
            print(key, {c: f"{sum(v)/len(v):.4e}" for c, v in sorted(cs.items())})
PY
grep "shape" $OUT/sq1.log

#!/bin/bash
# PMC passes over the dominant kernel alone (one rocprofv3 run per counter group; never combined with tracing
# domains other than --kernel-trace).  Output: gpurun_out/pmc_f43/<group>/... and a text summary.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
OUT=gpurun_out/pmc_f43
mkdir -p $OUT
rocprofv3 -L > $OUT/counters_list.txt 2>&1 || true
run() {  # name, counters...
  name=$1; shift
  timeout 300 rocprofv3 --kernel-trace --pmc "$@" -d $OUT/$name -o p --output-format csv -- python tools/f43_probe.py 3 > $OUT/$name.log 2>&1 || echo "pass $name failed"
}
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE
run sq2 SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVES
run sq3 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM
run sq4 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_THREAD_CYCLES_VALU
run tcp1 TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum
run ta1 TA_TA_BUSY_sum TA_BUFFER_WAVEFRONTS_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum
run tcc1 TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum
python - <<'PY'
import csv, glob, collections, os
out = open('gpurun_out/pmc_f43/summary.txt', 'w')
for d in sorted(glob.glob('gpurun_out/pmc_f43/*/')):
    for f in glob.glob(d + '**/*counter_collection.csv', recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            k = r['Kernel_Name']
            if 'f43_kernel' not in k: continue
            key = (k[:40], r.get('Grid_Size', '?'))
            acc[key][r['Counter_Name']].append(float(r['Counter_Value']))
        for key, cs in sorted(acc.items()):
            for c, v in sorted(cs.items()):
                line = f"{os.path.basename(os.path.dirname(d))} {key[0]} grid={key[1]} {c} n={len(v)} mean={sum(v)/len(v):.6e}"
                print(line); out.write(line + "\n")
PY
grep -c . gpurun_out/pmc_f43/counters_list.txt

#!/bin/bash
# PMC passes over the dominant kernel alone: memory-path stall counters, shipped build vs the no-halo-load probe build
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
OUT=gpurun_out/pmc_tlb
rm -rf $OUT; mkdir -p $OUT
rocprofv3 -L 2>/dev/null | grep -oE "(TCP|TA|TCC|SQ|UTCL|TCA)_[A-Z0-9_]+" | sort -u > $OUT/counters.txt
run() { name=$1; lib=$2; shift; shift
  FLOWSE_LIB_PATH=$lib PROBE_SHAPE=0 timeout 300 rocprofv3 --kernel-trace --pmc "$@" -d $OUT/$name -o p --output-format csv -- python tools/f43_probe.py 3 > $OUT/$name.log 2>&1 || echo "pass $name failed"; }
BASE=flowmse_amd/libflowse_hip.so
NOH=flowmse_amd/variants/nohalo/libflowse_hip.so
for v in base nohalo; do
  lib=$BASE; [ $v = nohalo ] && lib=$NOH
  run ${v}_a $lib TCP_PENDING_STALL_CYCLES TCP_TCC_READ_REQ_LATENCY TCP_TCC_READ_REQ TCP_UTCL1_TRANSLATION_MISS TCP_UTCL1_TRANSLATION_HIT TCP_UTCL1_REQUEST
  run ${v}_b $lib SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE
  run ${v}_c $lib TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES TA_DATA_STALLED_BY_TC_CYCLES TCP_TA_TCP_STATE_READ TCP_GATE_EN1 TCP_TCP_TA_DATA_STALL_CYCLES
done
python - <<'PY'
import csv, glob, collections
for d in sorted(glob.glob('gpurun_out/pmc_tlb/*/')):
    for f in glob.glob(d + '**/*counter_collection.csv', recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            k = r['Kernel_Name']
            if 'f43_kernel' not in k: continue
            acc[k[:40]][r['Counter_Name']].append(float(r['Counter_Value']))
        for key, cs in sorted(acc.items()):
            print(d.split('/')[-2], key, {c: f"{sum(v)/len(v):.4e}" for c, v in sorted(cs.items())})
PY
grep -h "failed\|rror" $OUT/*.log | head -5

#!/bin/bash
# L2 (TCC) hit / miss / memory-side request counters of conv3x3_w2d_kernel on ONE shape of tools/bench_wino.py (default index 0 =
# [8,256,256,128->128]): why does the kernel fetch 1.5x its algorithmic bytes from behind L2 although xcd_remap puts the channel
# blocks of a pixel tile on one XCD?  One rocprofv3 run per counter group, only --kernel-trace beside --pmc.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
IDX=${1:-0}
SHAPE_ARG=${SHAPE:+--shape $SHAPE}
[ -n "$SHAPE" ] && IDX=0
OUT=gpurun_out/pmc_w2d_tcc
rm -rf $OUT; mkdir -p $OUT
rocprofv3 -L 2>/dev/null | grep -o "TCC_[A-Z0-9_]*\|TCP_[A-Z0-9_]*" | sort -u > $OUT/counters.txt; wc -l $OUT/counters.txt
run() {  # name, counters...
  name=$1; shift
  timeout 300 rocprofv3 --kernel-trace --pmc "$@" -d $OUT/$name -o p --output-format csv -- python tools/bench_wino.py --gn --form w2d --only $IDX --iters 3 $SHAPE_ARG > $OUT/$name.log 2>&1 || echo "pass $name failed: $(tail -2 $OUT/$name.log)"
}
run t1 TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum
run t2 TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum
if [ -z "$SHORT" ]; then
run t3 TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_WRITE_REQ_sum
run t4 TCC_WRITE_sum TCC_WRITEBACK_sum TCC_NORMAL_EVICT_sum TCC_ALL_TC_OP_WB_WRITEBACK_sum
run t5 FETCH_SIZE
run t6 WRITE_SIZE
fi
python - <<'PY'
import csv, glob, collections, os
out = open('gpurun_out/pmc_w2d_tcc/summary.txt', 'w')
for d in sorted(glob.glob('gpurun_out/pmc_w2d_tcc/t*/')):
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

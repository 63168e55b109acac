#!/bin/bash
# HBM traffic of the dominant kernel, two settings side by side (separate --pmc passes as the guide prescribes)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
OUT=gpurun_out/traffic_ab; rm -rf $OUT; mkdir -p $OUT
CMD="python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-alt"
for v in tpb4 tpb1; do
  [ $v = tpb1 ] && export FLOWSE_F43_TPB1=1 || unset FLOWSE_F43_TPB1
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 200 rocprofv3 --kernel-trace --pmc $c -d $OUT/${v}_$c -o p --output-format csv -- $CMD > $OUT/${v}_$c.log 2>&1 || echo "$v $c failed"
  done
done
python - <<'PY'
import csv, glob, collections
for d in sorted(glob.glob('gpurun_out/traffic_ab/*/')):
    for f in glob.glob(d + '**/*counter_collection.csv', recursive=True):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if 'f43_kernel<2, false, 2>' in r['Kernel_Name']:
                acc[r['Counter_Name']].append(float(r['Counter_Value']))
        for c, v in acc.items(): print(d.split('/')[-2], c, len(v), sum(v)/len(v))
PY

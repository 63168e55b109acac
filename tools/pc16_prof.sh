# kernel duration of conv3x3_pc16_kernel alone (op context, shipped library) from rocprofv3, next to the s_memtime accounting
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
OUT=gpurun_out/pc16_prof; rm -rf $OUT; mkdir -p $OUT
for cfg in "8 128 0 128 256 256 1 1 1" "8 128 0 128 256 256 0 0 0"; do
  tag=$(echo $cfg | tr ' ' '_')
  timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/$tag -o p --output-format csv -- python tools/pc16_ts.py $cfg > $OUT/$tag.log 2>&1
  python - <<PY
import csv, glob
for f in glob.glob('$OUT/$tag/**/*kernel_stats.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'pc16' in r['Name'] or 'convert' in r['Name']: print('$cfg', r['Name'][:70], r['Calls'], 'avg us', float(r['AverageNs'])/1e3, 'min', float(r['MinNs'])/1e3, 'max', float(r['MaxNs'])/1e3)
PY
done

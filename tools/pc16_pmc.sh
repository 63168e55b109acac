# SQ counters of conv3x3_pc16_kernel alone (op context): effective clock (GRBM_GUI_ACTIVE / 8 XCDs / wall), MFMA-busy share
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
OUT=gpurun_out/pc16_pmc; rm -rf $OUT; mkdir -p $OUT
for cfg in "8 128 0 128 256 256 1 1 1" "8 128 0 128 256 256 0 0 0"; do
  tag=$(echo $cfg | tr ' ' '_')
  timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d $OUT/$tag -o p --output-format csv -- python tools/pc16_ts.py $cfg > $OUT/$tag.log 2>&1
  python - <<PY
import csv, glob, collections
dur = {}
for f in glob.glob('$OUT/$tag/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'pc16_kernel' in r['Kernel_Name']:
            dur.setdefault(r['Dispatch_Id'], int(r['End_Timestamp']) - int(r['Start_Timestamp']))
acc = collections.defaultdict(list)
for f in glob.glob('$OUT/$tag/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'pc16_kernel' in r['Kernel_Name']: acc[r['Counter_Name']].append(float(r['Counter_Value']))
d = sorted(dur.values()); wall = d[len(d)//2] if d else 0
print('$cfg', 'launches', len(d), 'median wall us', wall/1e3)
m = {k: sum(v)/len(v) for k, v in acc.items()}
for k, v in sorted(m.items()): print('   ', k, '%.4e' % v)
if wall and 'GRBM_GUI_ACTIVE' in m:
    clk = m['GRBM_GUI_ACTIVE'] / 8 / wall
    print('    effective clock GHz (GRBM_GUI_ACTIVE/8/wall)', round(clk, 3))
    if 'SQ_VALU_MFMA_BUSY_CYCLES' in m: print('    MFMA busy per SIMD', round(m['SQ_VALU_MFMA_BUSY_CYCLES'] / (1024 * clk * wall), 3))
    if 'SQ_WAVE_CYCLES' in m: print('    wave-cycles per wave (x4 quad) / wall -> GHz if resident all along', round(m['SQ_WAVE_CYCLES'] * 4 / 2048 / wall, 3))
PY
done

cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; OUT=gpurun_out/pmc16; mkdir -p $OUT
CMD="python bench.py --steps 1 --warmup 1 --precision bf16 --no-cpu-baseline --no-alt"
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/stats -o p --output-format csv -- $CMD > $OUT/stats.log 2>&1
run() { name=$1; shift; timeout 300 rocprofv3 --kernel-trace --pmc "$@" -d $OUT/$name -o p --output-format csv -- $CMD > $OUT/$name.log 2>&1 || echo "pass $name failed"; }
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE
run sq2 SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVES SQ_WAIT_INST_LDS
run sq4 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU
run tcp1 TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum
run fetch FETCH_SIZE
run write WRITE_SIZE
python - <<'PY'
import csv, glob, collections, os
for f in glob.glob('gpurun_out/pmc16/stats/**/*kernel_stats.csv', recursive=True):
    for i, r in enumerate(csv.DictReader(open(f))):
        if i < 12: print(r['Name'][:90], r['Calls'], r['AverageNs'], r['Percentage'])
for d in sorted(glob.glob('gpurun_out/pmc16/*/')):
    for f in glob.glob(d + '**/*counter_collection.csv', recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            k = r['Kernel_Name']
            if 'pc16' not in k and 'halo_bf16' not in k: continue
            acc[(k[:60], r.get('Grid_Size', '?'))][r['Counter_Name']].append(float(r['Counter_Value']))
        for key, cs in sorted(acc.items()):
            for c, v in sorted(cs.items()):
                print(os.path.basename(os.path.dirname(d)), key[0][:40], 'grid', key[1], c, 'n', len(v), 'mean %.5e' % (sum(v)/len(v)))
PY

# PMC counters of the sort kernels: bash tools/pmc_sort.sh [env assignments...]
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/pmc_sort; rm -rf $O; mkdir -p $O
cd /tmp
for v in "$@"; do export "$v"; done
timeout -k 10 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d $O/pmc1 -- python3 $GRAFT_REPO_ROOT/tools/sort_one.py 1e8 20 > /dev/null 2>&1
timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $O/pmc3 -- python3 $GRAFT_REPO_ROOT/tools/sort_one.py 1e8 20 > /dev/null 2>&1
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $GRAFT_REPO_ROOT/tools/sort_one.py 1e8 20 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv,glob,collections
for d in ('pmc1','pmc3'):
    agg=collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(f'gpurun_out/pmc_sort/{d}/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            n=r['Kernel_Name']
            if 'digit_s' in n: agg[n[:60]][r['Counter_Name']].append(float(r['Counter_Value']))
    for k,v in agg.items():
        print(d,k); print('   ', {c: round(sum(x)/len(x)/1e6,2) for c,x in v.items()})
for f in glob.glob('gpurun_out/pmc_sort/stats/**/*kernel_stats.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if any(k in r['Name'] for k in ('digit','scan_hist','transform')): print(r['Name'][:70], r['Calls'], round(float(r['AverageNs'])/1e3,1))
PY

# SQ PMC counters of every radix pass of one ORDER BY (per dispatch): bash tools/pmc_sort.sh [ENV=VALUE ...]   (SORT_WORKLOAD=sort20|sort32|sort64)
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/pmc_sort; rm -rf $O; mkdir -p $O
cd /tmp
for v in "$@"; do export "$v"; done
timeout -k 10 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $O/pmc1 -- python3 $GRAFT_REPO_ROOT/tools/op_one.py ${SORT_WORKLOAD:-sort20} > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv,glob,collections
for f in glob.glob('gpurun_out/pmc_sort/pmc1/**/*counter_collection.csv', recursive=True):
    per=collections.defaultdict(dict)
    for r in csv.DictReader(open(f)):
        if 'digit_scatter' in r['Kernel_Name']: per[int(r['Dispatch_Id'])][r['Counter_Name']]=float(r['Counter_Value'])/1e6
    for d in sorted(per)[-3:]:
        print(d, {k: round(v,1) for k,v in sorted(per[d].items())})
PY

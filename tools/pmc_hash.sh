# LDS / issue counters per dispatch of the hash GROUP BY consumer (and producer) of the sparse-key headline statement:
#   bash tools/pmc_hash.sh [scale]      -> gpurun_out/pmc_hash/summary.txt
# Three counter sets, each in its own run (with --kernel-trace only, as MI355X_MICROARCH.md prescribes).
export TMPDIR=/tmp
SCALE=${1:-0.5}
O=$GRAFT_REPO_ROOT/gpurun_out/pmc_hash; rm -rf $O; mkdir -p $O
cd /tmp
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --kernel-trace --output-format csv -d $O/pmc1 -- python3 $GRAFT_REPO_ROOT/tools/op_one.py sparse_gb $SCALE > $O/run1.log 2>&1
timeout -k 10 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM_RD --kernel-trace --output-format csv -d $O/pmc2 -- python3 $GRAFT_REPO_ROOT/tools/op_one.py sparse_gb $SCALE > $O/run2.log 2>&1
timeout -k 10 300 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_LDS_ATOMIC_RETURN SQ_LDS_MEM_VIOLATIONS GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc3 -- python3 $GRAFT_REPO_ROOT/tools/op_one.py sparse_gb $SCALE > $O/run3.log 2>&1
cd $GRAFT_REPO_ROOT
python - > $O/summary.txt <<'PY'
import csv,glob,collections
print("# SQ counters of the LAST dispatch of each fgb_* kernel of `tools/op_one.py sparse_gb` (counter values summed over the launch, in millions;")
print("# SQ cycle counters count per wave / per SIMD as the guide's table says; duration = End - Start of that dispatch in the counter run)")
for d in ("pmc1","pmc2","pmc3"):
    for f in glob.glob('gpurun_out/pmc_hash/%s/**/*counter_collection.csv' % d, recursive=True):
        per=collections.defaultdict(dict); name={}; dur={}
        for r in csv.DictReader(open(f)):
            if 'fgb_' in r['Kernel_Name']:
                k=int(r['Dispatch_Id']); per[k][r['Counter_Name']]=per[k].get(r['Counter_Name'],0.0)+float(r['Counter_Value'])/1e6
                nm=r['Kernel_Name']; nm=nm[nm.index('fgb_'):][:44]; name[k]=nm; dur[k]=(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3
        last={}
        for k in sorted(per): last[name[k]]=k
        for nm,k in last.items(): print(d, nm, "us=%.0f" % dur[k], {c: round(v,2) for c,v in sorted(per[k].items())})
PY
cat $O/summary.txt
find $O -name "*.csv" -size +5M -delete

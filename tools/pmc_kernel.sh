# SQ PMC counters per dispatch of the kernels whose name contains $1: bash tools/pmc_kernel.sh <kernel substring> <script.py> [args ...]   (e.g. jbucket op_one.py join_c4)
# (prints the last dispatch of every matching kernel; counters in millions, summed over the launch)
export TMPDIR=/tmp
PAT=$1; shift
O=$GRAFT_REPO_ROOT/gpurun_out/pmc_kernel; rm -rf $O; mkdir -p $O
cd /tmp
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $O/pmc1 -- python3 $GRAFT_REPO_ROOT/tools/$@ > /dev/null 2>&1
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc2 -- python3 $GRAFT_REPO_ROOT/tools/$@ > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
PAT="$PAT" python - <<'PY'
import csv,glob,collections,os
pat=os.environ["PAT"].split(",")
for d in ("pmc1","pmc2"):
    for f in glob.glob('gpurun_out/pmc_kernel/%s/**/*counter_collection.csv' % d, recursive=True):
        per=collections.defaultdict(dict); name={}
        for r in csv.DictReader(open(f)):
            if any(p in r['Kernel_Name'] for p in pat):
                per[int(r['Dispatch_Id'])][r['Counter_Name']]=float(r['Counter_Value'])/1e6; name[int(r['Dispatch_Id'])]=r['Kernel_Name'][:40]
        last={}
        for k in sorted(per): last[name[k]]=k
        for nm,k in last.items(): print(d, nm, {c: round(v,1) for c,v in sorted(per[k].items())})
PY
find gpurun_out/pmc_kernel -name "*.csv" -size +5M -delete

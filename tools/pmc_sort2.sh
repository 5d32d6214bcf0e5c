# HBM-side traffic of the sort kernels: bash tools/pmc_sort2.sh VARIANT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/pmc_sort2_$1; rm -rf $O; mkdir -p $O
cd /tmp
export HARK_SORT_TILED=$1
timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/f -- python3 $GRAFT_REPO_ROOT/tools/sort_one.py 1e8 20 > /dev/null 2>&1
timeout -k 10 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/w -- python3 $GRAFT_REPO_ROOT/tools/sort_one.py 1e8 20 > /dev/null 2>&1
timeout -k 10 200 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum --kernel-trace --output-format csv -d $O/t -- python3 $GRAFT_REPO_ROOT/tools/sort_one.py 1e8 20 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python - "$1" <<'PY'
import csv,glob,collections,sys
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(f'gpurun_out/pmc_sort2_{sys.argv[1]}/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        n=r['Kernel_Name']
        if 'digit_s' in n or 'digit_hist' in n: agg[n[:60]][r['Counter_Name']].append(float(r['Counter_Value']))
for k,v in agg.items():
    print('variant',sys.argv[1],k); print('   ', {c: round(sum(x)/len(x)/1e6,3) for c,x in v.items()}, '(1e6; FETCH/WRITE_SIZE in KiB -> GB: x1.024e-3; FETCH x2 on gfx950)')
PY

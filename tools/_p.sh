export TMPDIR=/tmp
O=gpurun_out/opsprof; rm -rf $O; mkdir -p $O
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 tools/ops_bench.py > $O/ops.log 2>&1
find $O -name "*kernel_trace.csv" -size +30M -delete
python tools/kstats.py $O/stats 40

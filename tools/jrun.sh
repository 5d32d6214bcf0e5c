# join timings + kernel breakdown of the three join workloads.  Usage (gpurun): bash tools/jrun.sh <outdir>
export TMPDIR=/tmp
O=$1; rm -rf $O; mkdir -p $O
timeout 200 rocprofv3 --kernel-trace --output-format csv -d $O/u32 -- python3 tools/join_one.py 1e8 1e7 > $O/u32.log 2>&1
timeout 200 rocprofv3 --kernel-trace --output-format csv -d $O/i64 -- python3 tools/join_one.py 1.25e8 1.25e7 i64 > $O/i64.log 2>&1
timeout 200 rocprofv3 --kernel-trace --output-format csv -d $O/c4 -- python3 tools/join_c4.py > $O/c4.log 2>&1
for w in u32 i64 c4; do grep join $O/$w.log | tail -2; python tools/jtrace.py $O/$w > $O/$w.trace; sed -n '/---- per/,$p' $O/$w.trace | head -14; done
find $O -name "*.csv" -size +5M -delete

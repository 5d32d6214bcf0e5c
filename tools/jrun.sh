# join timings + kernel breakdown of the join workloads.  Usage (gpurun): bash tools/jrun.sh <outdir>   (a fresh directory per run)
export TMPDIR=/tmp
O=$1; if [ -e "$O" ]; then echo "$O exists: name a fresh directory" >&2; exit 2; fi; mkdir -p $O
for w in join_u32 join_c4; do
  timeout 200 rocprofv3 --kernel-trace --output-format csv -d $O/$w -- python3 tools/op_one.py $w 1.0 4 > $O/$w.log 2>&1
  grep "$w:" $O/$w.log | tail -2; python tools/ktrace.py $O/$w 0.25 > $O/$w.trace; head -16 $O/$w.trace
done
find $O -name "*.csv" -size +5M -delete

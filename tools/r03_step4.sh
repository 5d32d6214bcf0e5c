export TMPDIR=/tmp
O=gpurun_out/r03s4; rm -rf $O; mkdir -p $O
for lib in libhark.so libhark_skew.so libhark_skew2.so libhark.so libhark_skew.so libhark_skew2.so; do
  HARK_LIB=$PWD/harkdb_amd/$lib timeout -k 10 200 python tools/alloc_probe.py plain 5 2 2>&1 | grep "plan [0-9]:" | sed "s/^/$lib /" >> $O/skew.log || exit 1
done
cat $O/skew.log

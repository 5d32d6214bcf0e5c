export TMPDIR=/tmp
O=gpurun_out/r03s3; rm -rf $O; mkdir -p $O
timeout -k 10 200 python tools/alloc_probe.py plain 6 2 > $O/plain.log 2>&1 && cat $O/plain.log &&
timeout -k 10 200 python tools/alloc_probe.py dummy_first 4 2 > $O/dummy_first.log 2>&1 && cat $O/dummy_first.log &&
timeout -k 10 200 python tools/alloc_probe.py dummy_mid 4 2 > $O/dummy_mid.log 2>&1 && cat $O/dummy_mid.log &&
timeout -k 10 200 python tools/alloc_probe.py plain 6 3 > $O/plain3.log 2>&1 && cat $O/plain3.log

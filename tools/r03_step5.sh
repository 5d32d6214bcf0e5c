export TMPDIR=/tmp
O=gpurun_out/r03s5; rm -rf $O; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_sharded.py tests/test_gpu_sharded2.py -x -q > $O/sharded.log 2>&1; echo "rc $?"; tail -40 $O/sharded.log

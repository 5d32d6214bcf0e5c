export TMPDIR=/tmp
O=gpurun_out/r03s10; rm -rf $O; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_sql.py tests/test_gpu_sharded.py tests/test_gpu_sharded2.py tests/test_gpu_hjoin.py -x -q > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -15 $O/pytest.log

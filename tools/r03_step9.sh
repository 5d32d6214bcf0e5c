export TMPDIR=/tmp
O=gpurun_out/r03s9; rm -rf $O; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_select.py tests/test_gpu_sql.py tests/test_gpu_property.py tests/test_gpu_segmented.py tests/test_gpu_groupby_join.py tests/test_gpu_hjoin.py -x -q > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -3 $O/pytest.log
for i in 1 2 3; do
timeout -k 10 120 python tools/c2_one.py 2>&1 | grep C2
done

export TMPDIR=/tmp
O=gpurun_out/r03s6; rm -rf $O; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_groupby_join.py tests/test_gpu_hjoin.py tests/test_gpu_sql.py tests/test_gpu_property.py tests/test_gpu_sharded.py -x -q > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -3 $O/pytest.log
for bits in 20 12 24; do
for i in 1 2; do
HARK_SORT_PLAN=byte timeout -k 10 120 python tools/sort_one.py 1e8 $bits 2>&1 | grep sort | tail -1 | sed "s/^/byte plan   /"
timeout -k 10 120 python tools/sort_one.py 1e8 $bits 2>&1 | grep sort | tail -1 | sed "s/^/range plan  /"
done; done

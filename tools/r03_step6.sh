export TMPDIR=/tmp
O=gpurun_out/r03s6; rm -rf $O; mkdir -p $O
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -4 $O/pytest.log
for i in 1 2; do
HARK_SORT_TILED=1 timeout -k 10 120 python tools/sort_one.py 1e8 20 2>&1 | grep sort | tail -1 | sed "s/^/old  /"
timeout -k 10 120 python tools/sort_one.py 1e8 20 2>&1 | grep sort | tail -1 | sed "s/^/new  /"
done

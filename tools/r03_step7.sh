export TMPDIR=/tmp
O=gpurun_out/r03s7; rm -rf $O; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_hjoin.py tests/test_gpu_groupby_join.py tests/test_gpu_sql.py tests/test_gpu_sharded.py tests/test_gpu_sharded2.py -x -q > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -4 $O/pytest.log
timeout -k 10 600 python -m pytest tests/test_gpu_fullsize.py -x -q -k "c4" > $O/pytest2.log 2>&1; echo "pytest rc $?"; tail -3 $O/pytest2.log
HARK_JOIN_NOCARRY=1 HARK_JOIN_NO_RANK_GATHER=1 timeout -k 10 200 python tools/join_c4.py 2>&1 | grep C4 | tail -2 | sed 's/^/before  /'
HARK_JOIN_NOCARRY=1 timeout -k 10 200 python tools/join_c4.py 2>&1 | grep C4 | tail -2 | sed 's/^/rank-gather only  /'
timeout -k 10 200 python tools/join_c4.py 2>&1 | grep C4 | tail -2 | sed 's/^/carry + rank-gather  /'
bash tools/jrun.sh $O/j > $O/jrun.log 2>&1; tail -48 $O/jrun.log

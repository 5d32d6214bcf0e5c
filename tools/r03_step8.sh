export TMPDIR=/tmp
O=gpurun_out/r03s8; rm -rf $O; mkdir -p $O
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -4 $O/pytest.log
timeout -k 10 200 python tools/join_c4.py 2>&1 | grep C4 | tail -2

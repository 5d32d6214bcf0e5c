set -x
export TMPDIR=/tmp
O=gpurun_out/r03s2; rm -rf $O; mkdir -p $O
timeout -k 10 300 python tools/fgb_ablate.py 1e9 "pairfmt=3;pairfmt=2;pairfmt=3,tag=1;pairfmt=2,tag=1;pairfmt=3,ablate=8192;pairfmt=3,ablate=12288" > $O/ablate.log 2>&1 && cat $O/ablate.log
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" | tee -a $O/pytest.log
tail -8 $O/pytest.log

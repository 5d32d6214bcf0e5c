timeout 600 python -m pytest tests/test_gpu_groupby_join.py -x -q 2>&1 | grep -E "passed|failed|Error|error|assert" | tail -8
timeout 300 python tools/join_one.py 2>&1 | grep join
timeout 300 python tools/ops_bench.py 2>&1 | grep -E "join"

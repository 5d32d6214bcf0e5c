timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|Error|error" | tail -5
timeout 300 python tools/ops_bench.py 2>&1 | grep -E "sort|join|sparse"

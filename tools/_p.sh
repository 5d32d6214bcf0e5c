timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|Error|error" | tail -5
timeout 300 python tools/fgb_ablate.py 1000000000 "pairfmt=1;pairfmt=2;pairfmt=2,ablate=4096" 1048576 2>&1 | tail -3
timeout 300 python tools/ops_bench.py 2>&1 | grep -E "query_groupby\(dense"
timeout 300 python tools/skew_bench.py 2>&1 | tail -4

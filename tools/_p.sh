timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|Error|error" | tail -5
timeout 300 python tools/fgb_ablate.py 1000000000 "ablate=0;ablate=4096;ablate=8192;ablate=12288" 1048576 2>&1 | tail -4
timeout 300 python tools/skew_bench.py 2>&1 | tail -6

"""ORDER BY of N rows by a u32 key with one carried column, a few times (for rocprofv3).  Usage: python tools/sort_one.py [rows] [key_bits]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from harkdb_amd.engine import Engine
N = int(float(sys.argv[1])) if len(sys.argv) > 1 else 100_000_000
bits = int(sys.argv[2]) if len(sys.argv) > 2 else 20
eng = Engine(0)
k, a = eng.alloc(N * 4), eng.alloc(N * 4)
eng.gen_columns(0x4861726B4442, 0, N, 1 << bits if bits < 32 else 0x7FFFFFFF, True, None, k, None)
eng.gen_columns(0x4861726B4442 + 9, 0, N, 1 << 16, True, None, a, None)
t = eng.table_from_device(N, [k, a], [np.uint32, np.uint32])
for r in range(4):
    eng.sync(); t0 = time.perf_counter(); res = eng.sort(t, 0, [0, 1]); eng.sync()
    print(f"sort {N} rows, {bits}-bit keys: {(time.perf_counter() - t0) * 1e3:.3f} ms", flush=True)
    res.free()

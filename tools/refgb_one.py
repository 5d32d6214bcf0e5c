"""The reference's own operator at size: query_groupby(db, 0, [1, 1], [sum, max]) over 1e8 rows, 2^20 dense u32 keys
(bench.py's REF_query_groupby_dense), a few times (for rocprofv3).  Usage: python tools/refgb_one.py [rows]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from harkdb_amd.engine import Engine
N = int(float(sys.argv[1])) if len(sys.argv) > 1 else 100_000_000
eng = Engine(0)
k, a = eng.alloc(N * 4), eng.alloc(N * 4)
eng.gen_columns(0x4861726B4442, 0, N, 1 << 20, True, None, k, None)
eng.gen_columns(0x4861726B4442 + 9, 0, N, 1 << 16, True, None, a, None)
t = eng.table_from_device(N, [k, a], [np.uint32, np.uint32])
for r in range(4):
    eng.sync(); t0 = time.perf_counter(); res = eng.query_groupby(t, 0, [1, 1], [2, 3]); eng.sync()
    print(f"query_groupby {N} rows: {(time.perf_counter() - t0) * 1e3:.3f} ms out={res.shape}", flush=True)
    res.free()

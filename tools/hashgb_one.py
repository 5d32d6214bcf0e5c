"""The reference group-by on SPARSE u32 keys (2^21 distinct keys spread over [0, 2^32)): hash partition + LDS hash tables.
A few times (for rocprofv3).  Usage: python tools/hashgb_one.py [rows] [distinct]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from harkdb_amd.engine import Engine
N = int(float(sys.argv[1])) if len(sys.argv) > 1 else 100_000_000
D = int(float(sys.argv[2])) if len(sys.argv) > 2 else 1 << 21
eng = Engine(0)
k, a = eng.alloc(N * 4), eng.alloc(N * 4)
eng.gen_columns(0x4861726B4442 + 9, 0, N, 1 << 16, True, None, a, None)
hk = (np.random.default_rng(1).integers(0, D, size=N, dtype=np.int64) * 2654435761 % (1 << 32)).astype(np.uint32)
eng.upload(k, hk)
t = eng.table_from_device(N, [k, a], [np.uint32, np.uint32])
for r in range(4):
    eng.sync(); t0 = time.perf_counter(); res = eng.query_groupby(t, 0, [1, 1], [2, 3]); eng.sync()
    print(f"query_groupby sparse {N} rows: {(time.perf_counter() - t0) * 1e3:.3f} ms out={res.shape}", flush=True)
    res.free()

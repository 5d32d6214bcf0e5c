import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from harkdb_amd.engine import Engine
N = 100_000_000
eng = Engine(0)
ks, a = eng.alloc(N * 4), eng.alloc(N * 4)
hk = (np.random.default_rng(1).integers(0, 1 << 21, size=N, dtype=np.int64) * 2654435761 % (1 << 32)).astype(np.uint32)
eng.upload(ks, hk)
eng.gen_columns(9, 0, N, 1 << 16, True, None, a, None)
t = eng.table_from_device(N, [ks, a], [np.uint32, np.uint32])
for r in range(3):
    t0 = time.perf_counter(); res = eng.query_groupby(t, 0, [1, 1], [2, 3]); eng.sync(); print((time.perf_counter() - t0) * 1e3, "ms", res.shape); res.free()

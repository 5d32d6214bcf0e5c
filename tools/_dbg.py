import os, sys, numpy as np
sys.path.insert(0, '/root/repo')
from harkdb_amd.engine import Engine
eng = Engine(0)
for seed in range(6):
    rng = np.random.default_rng(seed)
    n = 3_000_001
    key = (rng.integers(-2**62, 2**62, size=n)).astype(np.int64)
    t = eng.table_from_columns([key, np.arange(n, dtype=np.int32)])
    res = eng.sort(t, 0, [0, 1])
    res.free(); t.free()

import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from harkdb_amd.engine import Engine
eng = Engine(0)
rng = np.random.default_rng(1)
n = 3_000_001
pool = rng.integers(-2**62, 2**62, size=300)
u = rng.integers(-2**62, 2**62, size=n)
key = (rng.standard_normal(n) * 2.0**55).astype(np.int64)
t = eng.table_from_columns([key, np.arange(n, dtype=np.int32)])
eng.sort(t, 0, [0, 1]).free(); t.free()
ks = np.sort(key)
print("min", ks[0], "q", ks[[1, 10, 100, 1000, 1465, 3000]], "max", ks[-1], file=sys.stderr)

import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from harkdb_amd.engine import Engine
eng = Engine(0)
rng = np.random.default_rng(1)
for n in (3_000_001, 10_000_000, 30_000_000):
    key = (rng.standard_normal(n) * 2.0**55).astype(np.int64)
    print("n", n, file=sys.stderr, flush=True)
    t = eng.table_from_columns([key, np.arange(n, dtype=np.int32)])
    eng.sort(t, 0, [0, 1]).free(); t.free()

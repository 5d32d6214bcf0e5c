"""One statement of BASELINE configs[4] on one GPU's share, a few times (for rocprofv3): python tools/c5_one.py [rows]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from harkdb_amd import FutharkContext
N = int(float(sys.argv[1])) if len(sys.argv) > 1 else 500_000_000
G = 1 << 20
fc = FutharkContext()
eng = fc.FutEnv
SEED = 0x4861726B4442
cols = [eng.alloc(N * 4) for _ in range(4)]
key = eng.alloc(N * 4)
for j in range(0, 4, 2):
    eng.gen_columns(SEED + j, 0, N, G, False, cols[j], key if j == 0 else None, cols[j + 1])
fc.create_table_from_device("t", ["k", "c0", "c1", "c2", "c3"], [key] + cols, [np.int32] + [np.float32] * 4, N)
q = "select k, sum(c3), count(*), avg(c3) from t where c1 > 0.5 group by k having count(*) > 250 order by sum(c3) desc limit 10"
for r in range(4):
    eng.sync(); t0 = time.perf_counter(); out = fc.sql(q); ms = (time.perf_counter() - t0) * 1e3
    print(f"{ms:9.3f} ms out={out.shape}", flush=True)

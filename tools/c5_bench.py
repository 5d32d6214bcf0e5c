"""BASELINE configs[4] on one GPU's share: the full SELECT / WHERE / GROUP BY / HAVING / ORDER BY
pipeline through FutharkContext.sql() on a 16-column f32 table resident in HBM.
Usage: python tools/c5_bench.py [rows]   (default 5e8 rows = 32 GB of columns)"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from harkdb_amd import FutharkContext

N = int(float(sys.argv[1])) if len(sys.argv) > 1 else 500_000_000
G = 1 << 20
fc = FutharkContext()
eng = fc.FutEnv
SEED = 0x4861726B4442
cols = [eng.alloc(N * 4) for _ in range(16)]
key = eng.alloc(N * 4)
for j in range(0, 16, 2):
    eng.gen_columns(SEED + j, 0, N, G, False, cols[j], key if j == 0 else None, cols[j + 1])
schema = ["k"] + [f"c{j}" for j in range(16)]
fc.create_table_from_device("t", schema, [key] + cols, [np.int32] + [np.float32] * 16, N)
queries = [
    "select k, sum(c3), count(*) from t where c1 > 0.5 group by k",
    "select k, sum(c3), count(*), avg(c3) from t where c1 > 0.5 group by k having count(*) > 250 order by sum(c3) desc limit 10",
    "select k, sum(c3), max(c7), min(c9), count(*) from t where c1 > 0.5 group by k having count(*) > 250 order by sum(c3) desc limit 10",
    "select c0, c2 from t where c1 > 0.999 order by c0 limit 10",
]
for q in queries:
    ts = []
    for r in range(4):
        eng.sync(); t0 = time.perf_counter(); out = fc.sql(q); ts.append((time.perf_counter() - t0) * 1e3)
    ms = sorted(ts[1:])[1]
    print(f"{ms:9.3f} ms  {N / ms / 1e6:8.2f} Grows/s  out={out.shape}  {q}", flush=True)

"""min / max / avg of ONE column: the one-pass statistics path against separate passes (different columns).
Usage: python tools/stats_bench.py [rows] [groups]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from harkdb_amd import FutharkContext
N = int(float(sys.argv[1])) if len(sys.argv) > 1 else 200_000_000
G = int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 20
fc = FutharkContext(sql_mode=True)
eng = fc.FutEnv
cols = [eng.alloc(N * 4) for _ in range(6)]
eng.gen_columns(0x4861726B4442, 0, N, G, False, cols[0], cols[1], cols[2])       # p, k, x
eng.gen_columns(0x4861726B4442 + 5, 0, N, G, False, cols[3], None, cols[4])      # (unused), y
eng.gen_columns(0x4861726B4442 + 6, 0, N, G, False, None, None, cols[5])         # z
fc.create_table_from_device("t", ["p", "k", "x", "q", "y", "z"], cols, [np.float32, np.int32, np.float32, np.float32, np.float32, np.float32], N)
for stmt in ("select k, min(x), max(x), avg(x), count(*) from t where p > 0.5 group by k order by count(*) desc limit 5",
             "select k, min(x), max(y), avg(z), count(*) from t where p > 0.5 group by k order by count(*) desc limit 5",
             "select k, avg(x), count(*) from t where p > 0.5 group by k order by count(*) desc limit 5"):
    fc.sql_columns(stmt)
    ts = []
    for _ in range(5):
        eng.sync(); t0 = time.perf_counter(); fc.sql_columns(stmt); ts.append((time.perf_counter() - t0) * 1e3)
    print(f"{min(ts):8.3f} ms  {stmt}", flush=True)

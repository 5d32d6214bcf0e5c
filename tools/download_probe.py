import os, sys, time
import numpy as np
sys.path.insert(0, "/root/repo")
from harkdb_amd import FutharkContext
N = 200_000_000
G = 1 << 20
fc = FutharkContext()
eng = fc.FutEnv
SEED = 0x4861726B4442
cols = [eng.alloc(N * 4) for _ in range(4)]
key = eng.alloc(N * 4)
for j in range(0, 4, 2):
    eng.gen_columns(SEED + j, 0, N, G, False, cols[j], key if j == 0 else None, cols[j + 1])
fc.create_table_from_device("t", ["k"] + [f"c{j}" for j in range(4)], [key] + cols, [np.int32] + [np.float32] * 4, N)
q = "select k, sum(c3), count(*) from t where c1 > 0.5 group by k"
import cProfile, pstats
for r in range(3):
    eng.sync(); t0 = time.perf_counter(); out = fc.sql(q); print(f"sql(): {(time.perf_counter() - t0) * 1e3:.3f} ms", out.shape, out.dtype, flush=True)
for r in range(3):
    eng.sync(); t0 = time.perf_counter(); names, cs = fc.sql_columns(q); print(f"sql_columns(): {(time.perf_counter() - t0) * 1e3:.3f} ms", [c.dtype for c in cs], flush=True)
pr = cProfile.Profile(); pr.enable(); out = fc.sql(q); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(14)

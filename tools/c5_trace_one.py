"""ONE statement of BASELINE configs[4] (5e8 rows x (k, c1, c3, c7, c9)) a few times -- behind rocprofv3 --kernel-trace.
Usage: python tools/c5_trace_one.py [which: 0 pipeline, 1 three aggregates + LIMIT] [reps]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from harkdb_amd import FutharkContext

N, G = 500_000_000, 1 << 20
which = int(sys.argv[1]) if len(sys.argv) > 1 else 0
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
fc = FutharkContext()
eng = fc.FutEnv
SEED = 0x4861726B4442
c = [eng.alloc(N * 4) for _ in range(4)]
key = eng.alloc(N * 4)
eng.gen_columns(SEED, 0, N, G, False, c[0], key, c[1])
eng.gen_columns(SEED + 2, 0, N, G, False, c[2], None, c[3])
fc.create_table_from_device("t", ["k", "c1", "c3", "c7", "c9"], [key] + c, [np.int32] + [np.float32] * 4, N)
q = ["select k, sum(c3), count(*), avg(c3) from t where c1 > 0.5 group by k having count(*) > 250 order by sum(c3) desc limit 10",
     "select k, sum(c3), max(c7), min(c9), count(*) from t where c1 > 0.5 group by k having count(*) > 250 order by sum(c3) desc limit 10"][which]
for r in range(reps):
    eng.sync(); t0 = time.perf_counter(); out = fc.sql(q); ms = (time.perf_counter() - t0) * 1e3
    print(f"{ms:9.3f} ms  out={out.shape}", flush=True)

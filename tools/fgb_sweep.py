"""Sweep of the fused filter->group-by paths of libhark.so on one GPU.
Usage: python tools/fgb_sweep.py [N] ; prints ms, rows/s, algorithmic TB/s, fraction of 8 TB/s."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from harkdb_amd.engine import Engine, FgbPlan  # noqa: E402

N = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1 << 28
SEED = 0x4861726B4442
eng = Engine(0)
p, k, v = eng.alloc(N * 4), eng.alloc(N * 4), eng.alloc(N * 4)


def run(G, reps=5, **knobs):
    eng.gen_columns(SEED, 0, N, G, True, p, k, v)
    s, c = eng.alloc(G * 4), eng.alloc(G * 8)
    plan = FgbPlan(eng, N, G, **knobs)
    ts = []
    for r in range(reps + 1):
        plan.reset(); eng.sync()
        t0 = time.perf_counter()
        plan.run(p, ">", 0.5, k, v, N)
        plan.finish(s, c)
        ts.append((time.perf_counter() - t0) * 1e3)
    cnt = eng.download(c, G, np.int64)
    sm = eng.download(s, G, np.float32)
    ts = sorted(ts[1:])
    ms = ts[len(ts) // 2]
    byts = 12 * N + 16 * G
    print(f"G={G:8d} {knobs}: median {ms:8.3f} ms  min {ts[0]:8.3f} ms  {N / ms / 1e6:8.1f} Grows/s  "
          f"{byts / ms / 1e9:6.3f} TB/s  frac(8TB/s)={byts / ms / 1e9 / 8:5.3f}  survivors={cnt.sum()}  sum={sm.astype(np.float64).sum():.1f}",
          flush=True)
    plan.free(); eng.free(s); eng.free(c)


print(f"N = {N} rows ({N * 12 / 1e9:.2f} GB)")
for G in (16, 256, 4096, 8192):
    run(G, algo=1)
run(1 << 20, reps=2, algo=2)
for chunk in (1 << 26, 1 << 28, 1 << 30):
    if chunk <= N or chunk == 1 << 26:
        run(1 << 20, algo=3, chunk_rows=chunk)
run(1 << 20, algo=3, chunk_rows=1 << 28, shift=11)
run(1 << 20, algo=3, chunk_rows=1 << 28, shift=13)

"""Fused filter->group-by under key skew: every row in one key / Zipf-like hot keys (partition path)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from harkdb_amd.engine import Engine, FgbPlan
N = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1 << 24
G = 1 << 20
eng = Engine(0)
rng = np.random.default_rng(0)
p, k, v = eng.alloc(N * 4), eng.alloc(N * 4), eng.alloc(N * 4)
eng.upload(p, rng.random(N, dtype=np.float32)); eng.upload(v, rng.integers(0, 16, N).astype(np.float32))
s, c = eng.alloc(G * 4), eng.alloc(G * 8)
for name, keys in (("uniform", rng.integers(0, G, N)), ("one key", np.full(N, 5)), ("90% in 16 keys", np.where(rng.random(N) < 0.9, rng.integers(0, 16, N) * 4099, rng.integers(0, G, N))),
                   ("one bucket (4096 keys)", rng.integers(8192, 12288, N))):
    kk = keys.astype(np.int32)
    eng.upload(k, kk)
    plan = FgbPlan(eng, N, G, algo=3)
    ts = []
    for r in range(3):
        plan.reset(); eng.sync(); t0 = time.perf_counter(); plan.run(p, ">", 0.5, k, v, N); plan.finish(s, c); ts.append((time.perf_counter() - t0) * 1e3)
    cnt = eng.download(c, G, np.int64)
    print(f"{name:26s} {min(ts):9.3f} ms  {N / min(ts) / 1e6:8.2f} Grows/s  groups={int((cnt > 0).sum())} survivors={int(cnt.sum())}", flush=True)
    plan.free()

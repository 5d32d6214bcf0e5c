import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from harkdb_amd.engine import Engine, FgbPlan
N, G = 1 << 28, 1 << 20
eng = Engine(0)
p, k, v = eng.alloc(N * 4), eng.alloc(N * 4), eng.alloc(N * 4)
eng.gen_columns(0x4861726B4442, 0, N, G, True, p, k, v)
configs = []
for grid in (256, 512):
    for ab in (0, 1, 2, 6):
        configs.append((0, grid, 12, ab))
configs.append((0, 512, 13, 0))
for variant, grid, shift, ab in configs:
    plan = FgbPlan(eng, N, G, algo=3, chunk_rows=N, grid=grid, shift=shift)
    plan.set("variant", variant); plan.set("ablate", ab)
    ts = []
    for r in range(4):
        plan.reset(); eng.sync(); t0 = time.perf_counter(); plan.run(p, ">", 0.5, k, v, N); eng.sync(); ts.append((time.perf_counter() - t0) * 1e3)
    print(f"variant={variant} grid={grid} shift={shift} ablate={ab}: {min(ts[1:]):.3f} ms (producer+consumer)", flush=True)
    plan.free()

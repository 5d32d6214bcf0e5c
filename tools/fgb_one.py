"""Run one configuration of the fused filter->group-by a few times (for rocprofv3).
Usage: python tools/fgb_one.py N G [key=value ...]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from harkdb_amd.engine import Engine, FgbPlan  # noqa: E402

N, G = int(float(sys.argv[1])), int(float(sys.argv[2]))
knobs = {a.split("=")[0]: int(a.split("=")[1]) for a in sys.argv[3:]}
eng = Engine(0)
p, k, v = eng.alloc(N * 4), eng.alloc(N * 4), eng.alloc(N * 4)
eng.gen_columns(0x4861726B4442, 0, N, G, True, p, k, v)
s, c = eng.alloc(G * 4), eng.alloc(G * 8)
plan = FgbPlan(eng, N, G, **knobs)
for _ in range(5):
    plan.reset()
    plan.run(p, ">", 0.5, k, v, N)
    plan.finish(s, c)
print("survivors", eng.download(c, G, np.int64).sum())

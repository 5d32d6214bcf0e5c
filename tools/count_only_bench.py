"""COUNT-only (no value column) against SUM+COUNT on the headline workload.  Usage: python tools/count_only_bench.py [rows] [groups]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from harkdb_amd.engine import Engine, FgbPlan
N = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000_000
G = int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 20
eng = Engine(0)
p, k, v = eng.alloc(N * 4), eng.alloc(N * 4), eng.alloc(N * 4)
eng.gen_columns(0x4861726B4442, 0, N, G, True, p, k, v)
plan = FgbPlan(eng, N, G, timing=1)
for name, vv in (("SUM + COUNT", v), ("COUNT only ", None)):
    best = None
    for r in range(6):
        plan.reset(); plan.run(p, ">", 0.5, k, vv, N); ms, cnt = plan.timing()
        t = ms["producer"] + ms["single"] + ms["consumer"]
        best = t if r and (best is None or t < best) else best
        last = ms
    print(f"{name}: {best:.3f} ms  (producer/single {last['producer'] + last['single']:.3f}, consumer {last['consumer']:.3f})", flush=True)

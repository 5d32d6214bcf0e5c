"""Does the producer's time depend on WHERE its slab space (and the columns) were allocated?  Identical plans, created one
after the other in one process, timed interleaved.  Usage: python tools/alloc_probe.py MODE [nplans]
MODE: plain | dummy_first (24 GB allocated and kept before anything else) | dummy_mid (24 GB between the columns and the plans)"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from harkdb_amd.engine import Engine, FgbPlan

mode = sys.argv[1] if len(sys.argv) > 1 else "plain"
nplans = int(sys.argv[2]) if len(sys.argv) > 2 else 6
fmt = int(sys.argv[3]) if len(sys.argv) > 3 else 2
N, G = 10 ** 9, 1 << 20
eng = Engine(0)
keep = []
if mode == "dummy_first":
    keep.append(eng.alloc(24 << 30))
p, k, v = eng.alloc(N * 4), eng.alloc(N * 4), eng.alloc(N * 4)
eng.gen_columns(0x4861726B4442, 0, N, G, True, p, k, v)
if mode == "dummy_mid":
    keep.append(eng.alloc(24 << 30))
print(mode, "columns at", [hex(x) for x in (p, k, v)], "dummy at", [hex(x) for x in keep], flush=True)
plans = []
for i in range(nplans):
    pl = FgbPlan(eng, N, G, algo=3, pairfmt=fmt, timing=1)
    pl.reset(); pl.run(p, ">", 0.5, k, v, N); pl.timing()          # allocates the slab space now, in creation order
    probe = eng.alloc(4096); eng.free(probe)
    plans.append(pl)
    print("plan", i, "created; a 4 KiB block allocated after it sits at", hex(probe), flush=True)
res = [[] for _ in plans]
for r in range(7):
    for i, pl in enumerate(plans):
        pl.reset(); pl.run(p, ">", 0.5, k, v, N); ms, _ = pl.timing()
        if r:
            res[i].append((ms["producer"], ms["consumer"]))
for i, rr in enumerate(res):
    a = np.array(rr)
    print(f"{mode} fmt{fmt} plan {i}: producer min {a[:,0].min():.3f} med {np.median(a[:,0]):.3f}  consumer med {np.median(a[:,1]):.3f}", flush=True)

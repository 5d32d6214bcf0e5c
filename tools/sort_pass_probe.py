"""How much of a radix pass is the scatter pattern?  Same sort, keys whose digits take 2 / 16 / 256 values.
Usage: python tools/sort_pass_probe.py [rows]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from harkdb_amd.engine import Engine
N = int(float(sys.argv[1])) if len(sys.argv) > 1 else 100_000_000
eng = Engine(0)
rng = np.random.default_rng(0)
r = rng.integers(0, 2**32, size=N, dtype=np.uint64).astype(np.uint32)
pay = np.arange(N, dtype=np.uint32)
for name, keys in (("2 values per digit", (r & np.uint32(0x01010101))),
                   ("16 values per digit", (r & np.uint32(0x0F0F0F0F))),
                   ("256 values per digit", r)):
    t = eng.table_from_columns([keys, pay])
    ts = []
    for _ in range(4):
        eng.sync(); t0 = time.perf_counter(); res = eng.sort(t, 0, [1]); eng.sync(); ts.append((time.perf_counter() - t0) * 1e3); res.free()
    print(f"{name:22s} sort of {N} (key, payload) pairs, 4 passes: {min(ts):7.3f} ms", flush=True)
    t.free()

import os, sys, time
import numpy as np, torch
sys.path.insert(0, "/root/repo")
from harkdb_amd.engine import Engine
eng = Engine(0)
dev = torch.device("cuda", 0)
n = 100_000_000
g = torch.Generator(device=dev); g.manual_seed(1)
keys = torch.randint(-2**62, 2**62, (n,), dtype=torch.int64, device=dev, generator=g)
a = torch.arange(n, dtype=torch.int32, device=dev)
t = eng.table_from_device(n, [keys.data_ptr(), a.data_ptr()], [np.int64, np.int32], keepalive=(keys, a))
for r in range(4):
    eng.sync(); t0 = time.perf_counter(); res = eng.sort(t, 0, [0, 1]); eng.sync()
    print(f"ORDER BY i64 key, key + one column, {n} rows: {(time.perf_counter() - t0) * 1e3:.3f} ms", flush=True)
    res.free()

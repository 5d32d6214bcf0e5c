"""Every statement form that groups, sorts or joins by a key column, on the same 1e8-row table with its rows SHUFFLED and with the
table SORTED by that key (a table kept in key order): which forms still have a cliff?   python tools/statement_cluster_probe.py [rows]"""
import os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from harkdb_amd.engine import Engine
import bench

N = (int(float(sys.argv[1])) if len(sys.argv) > 1 else 10**8) // 4 * 4
G = 1 << 20
dev = torch.device("cuda", 0)
eng = Engine(0)
g = torch.Generator(device=dev); g.manual_seed(5)
p = torch.rand(N, device=dev, generator=g)
v = torch.randint(0, 16, (N,), device=dev, generator=g).to(torch.float32)
a = torch.randint(0, 1 << 16, (N,), device=dev, generator=g, dtype=torch.int32)
kd = torch.randint(0, G, (N,), device=dev, generator=g, dtype=torch.int32)
k64 = torch.randint(-2**62, 2**62, (N,), device=dev, generator=g, dtype=torch.int64)


def ms_of(fn, reps=3):
    ts = []
    for _ in range(reps + 1):
        eng.sync(); t0 = time.perf_counter(); r = fn(); eng.sync(); ts.append((time.perf_counter() - t0) * 1e3)
        rows = r.shape[0]; r.free()
    return min(ts[1:]), rows


for order in ("shuffled", "sorted"):
    k = kd if order == "shuffled" else kd.sort().values
    ks = (k.to(torch.int64) * bench.SPARSE_MUL).to(torch.int32)            # sparse keys in the SAME row order (ascending dense key: not ascending as i32, but clustered)
    ksort = ks if order == "shuffled" else ks.sort().values                 # ... and sorted as i32 (the typed entry's order)
    kusort = ks if order == "shuffled" else (ks ^ -(2**31)).sort().values ^ -(2**31)   # ... and as u32 (the reference entry's order)
    kl = k64 if order == "shuffled" else k64.sort().values
    torch.cuda.synchronize()                                               # (torch fills the columns on ITS stream)
    td = eng.table_from_device(N, [p.data_ptr(), k.data_ptr(), v.data_ptr(), a.data_ptr()], [np.float32, np.int32, np.float32, np.int32], keepalive=(p, k, v, a))
    ts = eng.table_from_device(N, [p.data_ptr(), ksort.data_ptr(), v.data_ptr()], [np.float32, np.int32, np.float32], keepalive=(p, ksort, v))
    tu = eng.table_from_device(N, [k.data_ptr(), a.data_ptr()], [np.uint32, np.uint32], keepalive=(k, a))
    th = eng.table_from_device(N, [kusort.data_ptr(), a.data_ptr()], [np.uint32, np.uint32], keepalive=(kusort, a))
    t64 = eng.table_from_device(N, [kl.data_ptr(), a.data_ptr()], [np.int64, np.int32], keepalive=(kl, a))
    W = [0.5]
    forms = [
        ("dense  SUM,COUNT WHERE p>0.5", lambda: eng.filter_groupby(td, [(0, ">", 0.5)], 1, [("sum", 2), ("count", 0)])),
        ("dense  SUM,COUNT (no WHERE)", lambda: eng.filter_groupby(td, [], 1, [("sum", 2), ("count", 0)])),
        ("dense  COUNT WHERE p>0.5", lambda: eng.filter_groupby(td, [(0, ">", 0.5)], 1, [("count", 0)])),
        ("dense  SUM,MAX,MIN,AVG,COUNT of one column", lambda: eng.filter_groupby(td, [(0, ">", 0.5)], 1, [("sum", 2), ("max", 2), ("min", 2), ("avg", 2), ("count", 0)])),
        ("dense  SUM(v),MAX(a) two columns", lambda: eng.filter_groupby(td, [(0, ">", 0.5)], 1, [("sum", 2), ("max", 3), ("count", 0)])),
        ("dense  MAX(a) i32", lambda: eng.filter_groupby(td, [(0, ">", 0.5)], 1, [("max", 3)])),
        ("sparse SUM,COUNT WHERE p>0.5", lambda: eng.filter_groupby(ts, [(0, ">", 0.5)], 1, [("sum", 2), ("count", 0)])),
        ("sparse five aggregates", lambda: eng.filter_groupby(ts, [(0, ">", 0.5)], 1, [("sum", 2), ("max", 2), ("min", 2), ("avg", 2), ("count", 0)])),
        ("reference query_groupby dense [sum,max]", lambda: eng.query_groupby(tu, 0, [1, 1], [2, 3])),
        ("reference query_groupby sparse [sum,max]", lambda: eng.query_groupby(th, 0, [1, 1], [2, 3])),
        ("ORDER BY u32 key (20 bits)", lambda: eng.sort(tu, 0, [0, 1])),
        ("ORDER BY i64 key", lambda: eng.sort(t64, 0, [0, 1])),
    ]
    for name, fn in forms:
        try:
            ms, rows = ms_of(fn)
            print(f"{order:9s} {name:46s} {ms:9.3f} ms  {rows:9d} rows  path {eng.last_groupby_path()}{' (window)' if eng.last_groupby_window() else ''}", flush=True)
        except Exception as e:                                              # noqa
            print(f"{order:9s} {name:46s} failed: {e}", flush=True)
    for t in (td, ts, tu, th, t64): t.free()

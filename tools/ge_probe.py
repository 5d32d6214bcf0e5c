import os, sys
import numpy as np, torch
sys.path.insert(0, '/root/repo')
from harkdb_amd.engine import Engine, FgbPlan
from harkdb_amd import dist as hd
import bench
N = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10**9
G = 1 << 20
dev = torch.device("cuda", 0)
eng = Engine(0); hd.share_stream(eng, dev)
p, k, v = (torch.empty(N, dtype=dt, device=dev) for dt in (torch.float32, torch.int32, torch.float32))
eng.gen_columns(bench.SEED, 0, N, G, True, p.data_ptr(), k.data_ptr(), v.data_ptr())
so, co = torch.empty(G, dtype=torch.float32, device=dev), torch.empty(G, dtype=torch.int64, device=dev)
print("rows with p == 0:", int((p == 0).sum()), "p < 0:", int((p < 0).sum()), "nan:", int(torch.isnan(p).sum()))
for cmp, thr in ((">=", 0.0), (">", -1.0), ("<", 2.0), (">=", 0.5), (">", 0.5), ("<=", 0.5), ("!=", 0.0), ("==", 0.0)):
    plan = FgbPlan(eng, N, G)
    plan.run(p.data_ptr(), cmp, thr, k.data_ptr(), v.data_ptr(), N)
    plan.finish(so.data_ptr(), co.data_ptr())
    keep = {">=": p >= thr, ">": p > thr, "<": p < thr, "<=": p <= thr, "!=": p != thr, "==": p == thr}[cmp]
    ref = torch.bincount(k[keep].to(torch.int64), minlength=G)
    bad = (ref != co).nonzero().flatten()
    print(cmp, thr, "hip", int(co.sum()), "torch", int(keep.sum()), "groups that differ", len(bad))
    if len(bad) and len(bad) < 100:
        g = int(bad[0]); rows = ((k == g) & keep).nonzero().flatten()
        print("  first differing group", g, "hip", int(co[g]), "torch", int(ref[g]), "p of its rows (min)", float(p[rows].min()), "rows with p==thr:", int((p[rows] == thr).sum()))
    plan.free()

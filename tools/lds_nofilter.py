import os, sys
import numpy as np, torch
sys.path.insert(0, '/root/repo')
from harkdb_amd.engine import Engine, FgbPlan
from harkdb_amd import dist as hd
import bench
N = 10**9
dev = torch.device("cuda", 0)
eng = Engine(0); hd.share_stream(eng, dev)
p, k, v = (torch.empty(N, dtype=dt, device=dev) for dt in (torch.float32, torch.int32, torch.float32))
for G in (4096, 13000):
    eng.gen_columns(bench.SEED, 0, N, G, True, p.data_ptr(), k.data_ptr(), v.data_ptr())
    so, co = torch.empty(G, dtype=torch.float32, device=dev), torch.empty(G, dtype=torch.int64, device=dev)
    for name, pp, cmp, thr in (("p = NULL", None, ">", 0.0), ("p >= 0", p.data_ptr(), ">=", 0.0), ("p > 0.5", p.data_ptr(), ">", 0.5), ("count(*) only, p = NULL", None, ">", 0.0)):
        plan = FgbPlan(eng, N, G, timing=1)
        vv = None if name.startswith("count") else v.data_ptr()
        def step():
            plan.reset(); plan.run(pp, cmp, thr, k.data_ptr(), vv, N); plan.finish(so.data_ptr(), co.data_ptr())
        ms = bench.event_ms(torch, step, warm=2, reps=7)
        bytes_ = (4 if pp else 0) + 4 + (4 if vv else 0)
        print(f"G={G} {name}: {ms:.3f} ms  {bytes_} B/row -> {bytes_ * N / ms / 1e9:.2f} TB/s = {bytes_ * N / ms / 1e9 / 8:.3f} of peak", flush=True)
        plan.free()

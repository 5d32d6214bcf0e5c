"""The headline statement at small G (single-pass LDS path), HIP-event timed.  Usage: [HARK_LIB=...] python tools/small_g.py [rows]"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from harkdb_amd.engine import Engine, FgbPlan
from harkdb_amd import dist as hd
N = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000_000
dev = torch.device("cuda", 0)
eng = Engine(0)
hd.share_stream(eng, dev)
p, k, v = (torch.empty(N, dtype=dt, device=dev) for dt in (torch.float32, torch.int32, torch.float32))
SEED = 0x4861726B4442
eng.gen_columns(SEED, 0, N, 16, True, p.data_ptr(), k.data_ptr(), v.data_ptr())
out = []
for G in (16, 4096, 13000):
    eng.gen_columns(SEED, 0, N, G, True, None, k.data_ptr(), None)
    so, co = torch.empty(G, dtype=torch.float32, device=dev), torch.empty(G, dtype=torch.int64, device=dev)
    plan = FgbPlan(eng, N, G)
    def step():
        plan.reset(); plan.run(p.data_ptr(), ">", 0.5, k.data_ptr(), v.data_ptr(), N); plan.finish(so.data_ptr(), co.data_ptr())
    for _ in range(3): step()
    torch.cuda.synchronize()
    ts = []
    for _ in range(15):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); step(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    ts.sort()
    out.append("G=%d %.3f ms (%.3f of peak)" % (G, ts[len(ts) // 2], 12.0 * N / (ts[len(ts) // 2] * 1e-3) / 8e12))
    plan.free()
print(os.environ.get("HARK_LIB", "libhark.so").split("/")[-1], " | ".join(out))

"""Does the relative placement of the three columns (and the partition workspace) in HBM move the producer?  One process,
one big allocation, columns carved at chosen byte offsets; interleaved rounds.  Usage: python tools/align_probe.py"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from harkdb_amd.engine import Engine, FgbPlan
from harkdb_amd import dist as hd
N, G = 1_000_000_000, 1 << 20
dev = torch.device("cuda", 0)
eng = Engine(0)
hd.share_stream(eng, dev)
big = torch.empty(3 * 4 * N + (64 << 20), dtype=torch.uint8, device=dev)
base = (big.data_ptr() + (2 << 20) - 1) // (2 << 20) * (2 << 20)          # 2 MiB aligned
SEED = 0x4861726B4442
plan = FgbPlan(eng, N, G, timing=1)
so, co = torch.empty(G, dtype=torch.float32, device=dev), torch.empty(G, dtype=torch.int64, device=dev)
col = 4 * N
col_al = (col + (2 << 20) - 1) // (2 << 20) * (2 << 20)                    # column pitch rounded to 2 MiB
configs = {
    "packed (pitch 4e9 B)": (0, col, 2 * col),
    "2 MiB-aligned pitch": (0, col_al, 2 * col_al),
    "aligned + 256 B skew": (0, col_al + 256, 2 * col_al + 512),
    "aligned + 4 KiB+256 skew": (0, col_al + 4352, 2 * col_al + 8704),
    "aligned + 64 KiB+4 KiB skew": (0, col_al + 69632, 2 * col_al + 139264),
    "aligned + 1 MiB+64 KiB skew": (0, col_al + 1114112, 2 * col_al + 2228224),
}
res = {k: [] for k in configs}
for rnd in range(4):
    for name, (o0, o1, o2) in configs.items():
        p, k, v = base + o0, base + o1, base + o2
        eng.gen_columns(SEED, 0, N, G, True, p, k, v)
        for it in range(3):
            plan.reset(); plan.run(p, ">", 0.5, k, v, N); plan.finish(so.data_ptr(), co.data_ptr())
        plan.timing()
        for it in range(5):
            plan.reset(); plan.run(p, ">", 0.5, k, v, N); plan.finish(so.data_ptr(), co.data_ptr())
        ms, cnt = plan.timing()
        res[name].append(ms["producer"] / cnt["producer"])
for name, v in res.items():
    print("%-32s producer %s  (min %.3f)" % (name, " ".join("%.3f" % x for x in v), min(v)), flush=True)

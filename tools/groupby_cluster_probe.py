"""The headline statement -- SELECT k, SUM(v), COUNT(*) FROM t WHERE p > 0.5 GROUP BY k, 2^20 groups -- when the key column is
SORTED or CLUSTERED (a table kept in key order): consecutive rows then fall into ONE of the producer's buckets, whose LDS rings
are sized for keys that scatter.   python tools/groupby_cluster_probe.py [rows] [shape ...]"""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from harkdb_amd.engine import Engine, FgbPlan
from harkdb_amd import dist as hd
import bench

N = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10**9
G = 1 << 20
dev = torch.device("cuda", 0)
eng = Engine(0)
hd.share_stream(eng, dev)
p, k, v = (torch.empty(N, dtype=dt, device=dev) for dt in (torch.float32, torch.int32, torch.float32))
eng.gen_columns(bench.SEED, 0, N, G, True, p.data_ptr(), k.data_ptr(), v.data_ptr())
so, co = torch.empty(G, dtype=torch.float32, device=dev), torch.empty(G, dtype=torch.int64, device=dev)
k0 = k.clone()
ref = None


def shape(name):
    if name.startswith("rot"):                                     # rotNN:SHAPE -- SHAPE with its rows regrouped so that the NN pieces of every batch of 4096 rows come from places
        parts, rest = int(name[3:name.index(":")]), name[name.index(":") + 1:]   # n / NN rows apart (what a producer whose waves read different batches would see)
        shape(rest)
        piece = 4096 // parts
        nb = N // 4096
        S = (nb // parts) | 1
        idx = (torch.arange(nb, device=dev)[:, None] + torch.arange(parts, device=dev)[None, :] * S) % nb
        kk = k[:nb * 4096].view(nb, parts, piece)
        k[:nb * 4096] = kk[idx, torch.arange(parts, device=dev)[None, :]].reshape(-1)
        return
    if name == "random": k.copy_(k0); return
    if name == "sorted": k.copy_(k0.sort().values); return
    if name == "descending": k.copy_(k0.sort(descending=True).values); return
    if name.startswith("noisy"):                                  # sorted, then NN per cent of the rows overwritten with keys from anywhere (late rows in a table kept in key order)
        k.copy_(k0.sort().values)
        m = int(N * float(name[5:]) / 100)
        idx = torch.randint(0, N, (m,), device=dev)
        k[idx] = k0[idx]
        return
    w = int(name[6:] if name.startswith("blocks") else name[4:])
    m = N // w
    if name.startswith("blocks"):                                  # sorted inside blocks of w rows
        k.copy_(k0); k[:m * w] = k0[:m * w].view(m, w).sort(dim=1).values.reshape(-1); return
    s = k0.sort().values                                           # runs of w sorted rows, shuffled as wholes
    k.copy_(s); k[:m * w] = s[:m * w].view(m, w)[torch.randperm(m, device=dev)].reshape(-1)


SHAPES = sys.argv[2:] or ["random", "sorted", "descending", "noisy0.1", "noisy1", "noisy5", "noisy20", "blocks1000000", "blocks65536", "blocks8192", "runs4096", "runs256", "runs16"]
for name in SHAPES:
    shape(name)
    torch.cuda.synchronize()
    for label, pp in (("p > 0.5", p.data_ptr()), ("p = NULL", None)):
        plan = FgbPlan(eng, N, G, timing=1)

        def step():
            plan.reset()
            plan.run(pp, ">", 0.5, k.data_ptr(), v.data_ptr(), N)
            plan.finish(so.data_ptr(), co.data_ptr())

        ms = bench.event_ms(torch, step, warm=2, reps=5)
        kms, kl = plan.timing()
        tot = int(co.sum().item())
        print(f"{name:14s} {label}: step {ms:7.3f} ms  producer {kms['producer'] / max(1, kl['producer']):7.3f}  consumer {kms['consumer'] / max(1, kl['consumer']):6.3f}  rows counted {tot}", flush=True)
        plan.free()

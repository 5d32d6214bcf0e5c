"""What a probe column that is SORTED or CLUSTERED by the join key costs (a fact table kept in key order, the output of a GROUP BY
joined back): consecutive rows then fall into the same bucket of the partition, whose rings and workgroup-private slabs are sized
for rows that scatter.  1e8 probe rows against 1e7 unique build keys (u32), every probe key drawn from [0, 2^30) as in
tools/join_skew_probe.py (~1 % of the rows find a partner), or from the build keys themselves (`match`: every row does).
Usage: python tools/join_cluster_probe.py            (the table of profiles/r06_join_cluster.txt)
       python tools/join_cluster_probe.py CASE ...   (those shapes, three times each; HARK_JOIN_CLUSTERED=0 / 1 forces either path)"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from harkdb_amd.engine import Engine
eng = Engine(0); dev = torch.device("cuda", 0)
n, m = 100_000_000, 10_000_000
g = torch.Generator(device=dev); g.manual_seed(11)
I64 = bool(os.environ.get("PROBE_I64"))                               # PROBE_I64=1: the same shapes on i64 keys (x 2^20 + 5: beyond 32 bits)
build = (torch.arange(m, device=dev, dtype=torch.int64) * 107 % (1 << 30)).to(torch.int32)
bval = torch.randint(0, 1 << 16, (m,), dtype=torch.int32, device=dev, generator=g)
pval = torch.randint(0, 1 << 16, (n,), dtype=torch.int32, device=dev, generator=g)
KDT = np.int64 if I64 else np.uint32
wide = lambda t: (t.to(torch.int64) << 20) + 5 if I64 else t
build_k = wide(build)
tb = eng.table_from_device(m, [build_k.data_ptr(), bval.data_ptr()], [KDT, np.uint32], keepalive=(build_k, bval))


def shape(name):
    if name.startswith("match"):
        probe = build[torch.randint(0, m, (n,), device=dev, generator=g)]
        name = name[len("match_"):] or "random"
    else:
        probe = torch.randint(0, 1 << 30, (n,), dtype=torch.int32, device=dev, generator=g)
    if name == "random": return probe
    if name == "sorted": return probe.sort().values
    if name == "descending": return probe.sort(descending=True).values
    if name.startswith("noisy"):                                    # sorted, then NN per cent of the rows overwritten with keys from anywhere
        s = probe.sort().values
        idx = torch.randint(0, n, (int(n * float(name[5:]) / 100),), device=dev, generator=g)
        s[idx] = probe[idx]
        return s
    if name.startswith("blocks"):                                   # sorted inside blocks of that many rows
        w = int(name[6:])
        k = n // w
        return torch.cat([probe[:k * w].view(k, w).sort(dim=1).values.reshape(-1), probe[k * w:].sort().values]).contiguous()
    if name.startswith("runs"):                                     # sorted, then the runs of that many rows shuffled as wholes
        w = int(name[4:])
        s = probe.sort().values
        k = n // w
        head = s[:k * w].view(k, w)
        return torch.cat([head[torch.randperm(k, device=dev, generator=g)].reshape(-1), s[k * w:]]).contiguous()
    raise SystemExit(f"unknown shape {name}")


CASES = ("random", "sorted", "descending", "noisy1", "noisy20", "blocks1000000", "blocks100000", "blocks30000", "blocks10000", "runs4096", "runs256", "runs16", "match_random", "match_sorted", "match_runs4096")
if len(sys.argv) > 1: CASES = tuple(sys.argv[1:])
for case in CASES:
    probe = wide(shape(case))
    torch.cuda.synchronize()
    tp = eng.table_from_device(n, [probe.data_ptr(), pval.data_ptr()], [KDT, np.uint32], keepalive=(probe, pval))
    ts = []
    for r in range(3):
        eng.sync(); t0 = time.perf_counter(); res = eng.join(tp, tb, 0, 0, [0, 1], [1]); eng.sync(); ts.append((time.perf_counter() - t0) * 1e3); rows = res.shape[0]; res.free()
    print(f"{case:16s}: {min(ts):7.3f} ms, {rows} result rows, path: {eng.last_join_path()}", flush=True)
    tp.free(); del probe

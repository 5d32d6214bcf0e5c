"""How skewed may a foreign-key join be before the partitioned join gives up and the sort-merge path runs?  (the survivor
bins of a bucket are sized before the partition runs; a bin that draws more than 2-4 x its even share sends the WHOLE join away.)
1e8 probe rows against 1e7 unique build keys (u32); a share `hot` of the probe rows is redirected to `nhot` build keys that exist.
Usage: python tools/join_skew_probe.py            (the table of profiles/r06_join_skew.txt)
       python tools/join_skew_probe.py HOT NHOT   (one shape four times, for a kernel trace: tools/ktrace.py DIR 0.25)"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from harkdb_amd.engine import Engine
eng = Engine(0); dev = torch.device("cuda", 0)
n, m = 100_000_000, 10_000_000
g = torch.Generator(device=dev); g.manual_seed(7)
build = torch.randperm(1 << 30, device=dev, generator=g)[:m].to(torch.int32) if False else (torch.arange(m, device=dev, dtype=torch.int64) * 107 % (1 << 30)).to(torch.int32)
bval = torch.randint(0, 1 << 16, (m,), dtype=torch.int32, device=dev, generator=g)
pval = torch.randint(0, 1 << 16, (n,), dtype=torch.int32, device=dev, generator=g)
tb = eng.table_from_device(m, [build.data_ptr(), bval.data_ptr()], [np.uint32, np.uint32], keepalive=(build, bval))
CASES = ((0.0, 1), (0.0001, 1), (0.0002, 1), (0.0005, 1), (0.001, 1), (0.01, 1), (0.05, 1), (0.001, 100), (0.01, 100), (0.1, 100), (0.3, 10_000), (0.5, 100_000))
if len(sys.argv) > 2: CASES = ((float(sys.argv[1]), int(sys.argv[2])),)
for hot, nhot in CASES:
    probe = torch.randint(0, 1 << 30, (n,), dtype=torch.int32, device=dev, generator=g)          # ~1 % of these exist in the build side
    k = int(n * hot)
    if k:
        idx = torch.randint(0, n, (k,), device=dev, generator=g)
        probe[idx] = build[torch.randint(0, nhot, (k,), device=dev, generator=g)]
    torch.cuda.synchronize()
    tp = eng.table_from_device(n, [probe.data_ptr(), pval.data_ptr()], [np.uint32, np.uint32], keepalive=(probe, pval))
    ts = []
    for r in range(4 if len(sys.argv) > 2 else 3):
        eng.sync(); t0 = time.perf_counter(); res = eng.join(tp, tb, 0, 0, [0, 1], [1]); eng.sync(); ts.append((time.perf_counter() - t0) * 1e3); rows = res.shape[0]; res.free()
    print(f"{hot * 100:6.2f} % of the probe rows on {nhot:6d} hot keys: {min(ts):7.3f} ms, {rows} result rows", flush=True)
    tp.free(); del probe

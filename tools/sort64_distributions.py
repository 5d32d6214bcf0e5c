"""ORDER BY an i64 key + one column for several sizes and key distributions, with and without the three-sweep sort in front
(HARK_SORT_NO_MSD=1): python tools/sort64_distributions.py"""
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from harkdb_amd.engine import Engine
eng = Engine(0); dev = torch.device("cuda", 0)
g = torch.Generator(device=dev); g.manual_seed(3)
for n in (3_000_000, 30_000_000, 100_000_000, 200_000_000):
    au = torch.randint(0, 1 << 16, (n,), dtype=torch.int32, device=dev, generator=g)
    for kind in ("uniform", "normal", "exp", "sorted", "dups10"):
        if kind == "uniform": k = torch.randint(-2**62, 2**62, (n,), dtype=torch.int64, device=dev, generator=g)
        elif kind == "normal": k = (torch.randn(n, device=dev, generator=g, dtype=torch.float64) * 2.0**55).to(torch.int64)
        elif kind == "exp": k = (-torch.log(torch.rand(n, device=dev, generator=g, dtype=torch.float64)) * 2.0**52).to(torch.int64)
        elif kind == "sorted": k = torch.sort(torch.randint(-2**62, 2**62, (n,), dtype=torch.int64, device=dev, generator=g))[0]
        else: k = torch.randint(0, n // 10, (n,), dtype=torch.int64, device=dev, generator=g) * (2**62 // (n // 10))
        torch.cuda.synchronize()
        t = eng.table_from_device(n, [k.data_ptr(), au.data_ptr()], [np.int64, np.int32], keepalive=(k, au))
        out = []
        for env in ("", "1"):
            if env: os.environ["HARK_SORT_NO_MSD"] = "1"
            else: os.environ.pop("HARK_SORT_NO_MSD", None)
            ts = []
            for r in range(3):
                eng.sync(); t0 = time.perf_counter(); res = eng.sort(t, 0, [0, 1]); eng.sync(); ts.append((time.perf_counter() - t0) * 1e3); res.free()
            out.append(min(ts))
        print(f"n={n:>11,d} {kind:8s} three sweeps first {out[0]:8.3f} ms   tuple passes first {out[1]:8.3f} ms", flush=True)
        t.free(); del k
    del au

import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from harkdb_amd.engine import Engine
eng = Engine(0); dev = torch.device("cuda", 0)
n = 100_000_000
g = torch.Generator(device=dev); g.manual_seed(3)
au = torch.randint(0, 1 << 16, (n,), dtype=torch.int32, device=dev, generator=g)
def keys(kind):
    if kind == "uniform": return torch.randint(-2**62, 2**62, (n,), dtype=torch.int64, device=dev, generator=g)
    if kind == "normal": return (torch.randn(n, device=dev, generator=g, dtype=torch.float64) * 2.0**55).to(torch.int64)
    if kind == "dups_1e5": return torch.randint(0, 100_000, (n,), dtype=torch.int64, device=dev, generator=g) * 92233720368547
    if kind == "dups_1e7": return torch.randint(0, 10_000_000, (n,), dtype=torch.int64, device=dev, generator=g) * 922337203685
    if kind == "exp": return (-torch.log(torch.rand(n, device=dev, generator=g, dtype=torch.float64)) * 2.0**52).to(torch.int64)
    if kind == "sorted": return torch.sort(torch.randint(-2**62, 2**62, (n,), dtype=torch.int64, device=dev, generator=g))[0]
for kind in ("uniform", "normal", "dups_1e5", "dups_1e7", "sorted", "exp"):
    k = keys(kind); torch.cuda.synchronize()
    t = eng.table_from_device(n, [k.data_ptr(), au.data_ptr()], [np.int64, np.int32], keepalive=(k, au))
    for env in ("", "1"):
        if env: os.environ["HARK_SORT_NO_MSD"] = "1"
        else: os.environ.pop("HARK_SORT_NO_MSD", None)
        ts = []
        for r in range(3):
            eng.sync(); t0 = time.perf_counter(); res = eng.sort(t, 0, [0, 1]); eng.sync(); ts.append((time.perf_counter() - t0) * 1e3); res.free()
        print(f"{kind:10s} {'tuple passes' if env else 'msd first   '} {min(ts):7.3f} ms", flush=True)
    t.free(); del k

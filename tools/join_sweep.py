"""Join time over sizes, partitioned path against the sort-merge path (HARK_JOIN_SORTMERGE=1 in a child process): where
does the partitioned path start to pay?  Usage: python tools/join_sweep.py [u32|i64]"""
import os, subprocess, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
kind = sys.argv[1] if len(sys.argv) > 1 else "u32"
if len(sys.argv) > 2:                      # child: one size
    from harkdb_amd.engine import Engine
    n, s = int(sys.argv[2]), int(sys.argv[3])
    eng = Engine(0)
    rng = np.random.default_rng(1)
    dt = np.uint32 if kind == "u32" else np.int64
    info = np.iinfo(dt)
    rk = np.unique(rng.integers(info.min, info.max, size=s + s // 8, dtype=np.int64).astype(dt))[:s]
    rk = rk[rng.permutation(len(rk))]
    lk = rng.integers(info.min, info.max, size=n, dtype=np.int64).astype(dt)
    hit = rng.random(n) < 0.5
    lk[hit] = rk[rng.integers(0, len(rk), size=int(hit.sum()))]
    t1 = eng.table_from_columns([lk, np.arange(n, dtype=np.int32)])
    t2 = eng.table_from_columns([rk, np.arange(len(rk), dtype=np.int32)])
    ts = []
    for r in range(5):
        eng.sync(); t0 = time.perf_counter(); res = eng.join(t1, t2, 0, 0, [1], [1]); eng.sync(); ts.append((time.perf_counter() - t0) * 1e3); res.free()
    print(f"{sorted(ts[1:])[1]:.3f}")
    sys.exit(0)
for n in (1 << 17, 1 << 18, 1 << 19, 1 << 20, 1 << 22, 1 << 24):
    for s in (max(4096, n // 64), n // 8):
        row = []
        for env in ({}, {"HARK_JOIN_SORTMERGE": "1"}):
            out = subprocess.run([sys.executable, __file__, kind, str(n), str(s)], capture_output=True, text=True, env=dict(os.environ, **env), timeout=300)
            row.append(out.stdout.strip().splitlines()[-1] if out.returncode == 0 and out.stdout.strip() else "fail")
        print(f"{kind} n={n:9d} s={s:8d}  partitioned {row[0]:>8s} ms   sort-merge {row[1]:>8s} ms", flush=True)

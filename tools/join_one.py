"""One join configuration a few times (for rocprofv3).  Usage: python tools/join_one.py [probe_rows] [build_rows] [u32|i64]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from harkdb_amd.engine import Engine
N = int(float(sys.argv[1])) if len(sys.argv) > 1 else 100_000_000
M = int(float(sys.argv[2])) if len(sys.argv) > 2 else N // 10
eng = Engine(0)
SEED = 0x4861726B4442
kb, vb, kp, a = eng.alloc(M * 4), eng.alloc(M * 4), eng.alloc(N * 4), eng.alloc(N * 4)
eng.gen_columns(SEED + 3, 0, M, 1 << 30, True, None, kb, None)
eng.gen_columns(SEED + 4, 0, M, 1 << 16, True, None, vb, None)
eng.gen_columns(SEED + 5, 0, N, 1 << 30, True, None, kp, None)
eng.gen_columns(SEED + 9, 0, N, 1 << 16, True, None, a, None)
WIDE = len(sys.argv) > 3 and sys.argv[3] == "i64"
if WIDE:                                   # BASELINE configs[3]: i64 keys (here: the u32 keys spread over 64 bits on the host)
    def widen(ptr, n):
        k32 = eng.download(ptr, n, np.uint32).astype(np.uint64)
        k64 = ((k32 * np.uint64(0x9E3779B97F4A7C15)) >> np.uint64(1)).astype(np.int64) - np.int64(1 << 62)
        q = eng.alloc(n * 8); eng.upload(q, k64); return q
    kp, kb = widen(kp, N), widen(kb, M)
kdt = np.int64 if WIDE else np.uint32
tp = eng.table_from_device(N, [kp, a], [kdt, np.uint32])
tb = eng.table_from_device(M, [kb, vb], [kdt, np.uint32])
for r in range(4):
    eng.sync(); t0 = time.perf_counter(); res = eng.join(tp, tb, 0, 0, [0, 1], [1]); eng.sync()
    print(f"join {N} x {M}: {(time.perf_counter() - t0) * 1e3:.3f} ms out={res.shape}", flush=True)
    res.free()

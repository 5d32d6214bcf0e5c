"""Wall-clock timings of the other BASELINE configs' operators on one GPU (entries include result
allocation and the host sync they need).  Usage: python tools/ops_bench.py [rows]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from harkdb_amd.engine import Engine

N = int(float(sys.argv[1])) if len(sys.argv) > 1 else 100_000_000
eng = Engine(0)
SEED = 0x4861726B4442


def timeit(name, fn, bytes_alg, reps=5):
    """bytes_alg: algorithmic bytes, or a function of the result's shape."""
    ts = []
    for r in range(reps + 1):
        eng.sync(); t0 = time.perf_counter(); res = fn(); eng.sync(); ts.append((time.perf_counter() - t0) * 1e3)
        shape = res.shape
        res.free()
    ts = sorted(ts[1:])
    ms = ts[len(ts) // 2]
    if callable(bytes_alg):
        bytes_alg = bytes_alg(shape)
    print(f"{name:58s} {ms:9.3f} ms  {N / ms / 1e6:8.2f} Grows/s  alg {bytes_alg / ms / 1e9:6.3f} TB/s ({bytes_alg / ms / 1e9 / 8:5.3f} of peak)  out={shape}", flush=True)


# C2: 8 f32 columns, SELECT c0, c2 WHERE c1 > 0.5
cols = [eng.alloc(N * 4) for _ in range(8)]
for j in range(0, 8, 2):
    eng.gen_columns(SEED + j, 0, N, 1 << 20, False, cols[j], None, cols[j + 1])
t8 = eng.table_from_device(N, cols, [np.float32] * 8)
timeit("C2  select c0,c2 where c1>0.5 (f32, 8 cols)", lambda: eng.filter_sel(t8, 1, ">", 0.5, [0, 2], want_row_index=True), lambda sh: 12 * N + 16 * sh[0])   # bench.py's C2 model: 3 columns read, 2 x f32 + i64 row index per survivor
timeit("C2' select c0,c2 where c1>0.5, no row index", lambda: eng.filter_sel(t8, 1, ">", 0.5, [0, 2], want_row_index=False), lambda sh: 12 * N + 8 * sh[0])
timeit("C1' projection select c0,c2 (query_sel)", lambda: eng.query_sel(t8, [0, 2]), 16 * N)

# reference-semantics group-by / sort / join on u32 columns
k = eng.alloc(N * 4); a = eng.alloc(N * 4)
eng.gen_columns(SEED, 0, N, 1 << 20, True, None, k, None)
eng.gen_columns(SEED + 9, 0, N, 1 << 16, True, None, a, None)
tu = eng.table_from_device(N, [k, a], [np.uint32, np.uint32])
# one fresh device buffer + table per key distribution (column statistics are cached with a table: tables are immutable),
# and every line names the path that actually ran (hark_context_last_groupby_path) instead of assuming it
def gb_line(label, table):
    timeit(label, lambda: eng.query_groupby(table, 0, [1, 1], [2, 3]), 12 * N)
    print(f"    ^ path taken: {eng.last_groupby_path()}", flush=True)


gb_line("query_groupby(dense key, 2^20 groups; sum, max)", tu)
ks = eng.alloc(N * 4)
eng.gen_columns(SEED + 11, 0, N, 1 << 31, True, None, ks, None)
tsparse = eng.table_from_device(N, [ks, a], [np.uint32, np.uint32])
gb_line("query_groupby(~1e8 distinct sparse keys; sum, max)", tsparse)
tsparse.free(); eng.free(ks)
kh = eng.alloc(N * 4)
hk = (np.random.default_rng(1).integers(0, 1 << 21, size=N, dtype=np.int64) * 2654435761 % (1 << 32)).astype(np.uint32)
eng.upload(kh, hk)
thash = eng.table_from_device(N, [kh, a], [np.uint32, np.uint32])
gb_line("query_groupby(2^21 distinct keys spread over [0,2^32); sum, max)", thash)
thash.free(); eng.free(kh)
timeit("sort by u32 key, 2 columns", lambda: eng.sort(tu, 0, [0, 1]), 16 * N)
M = N // 10
kb = eng.alloc(M * 4); vb = eng.alloc(M * 4)
eng.gen_columns(SEED + 3, 0, M, 1 << 30, True, None, kb, None)
eng.gen_columns(SEED + 4, 0, M, 1 << 16, True, None, vb, None)
kp = eng.alloc(N * 4)
eng.gen_columns(SEED + 5, 0, N, 1 << 30, True, None, kp, None)
tp = eng.table_from_device(N, [kp, a], [np.uint32, np.uint32])
tb = eng.table_from_device(M, [kb, vb], [np.uint32, np.uint32])
timeit(f"join probe {N} x build {M} on u32 key (partitioned)", lambda: eng.join(tp, tb, 0, 0, [0, 1], [1]), 12 * (N + M))

"""Ingest rate: host numpy columns -> device table (hark_table_new_columns), and result download."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from harkdb_amd.engine import Engine
eng = Engine(0)
n = 256 << 20                      # 3 columns x 1 GiB
cols = [np.random.default_rng(i).random(n, dtype=np.float32) for i in range(3)]
for r in range(3):
    t0 = time.perf_counter(); t = eng.table_from_columns(cols); eng.sync(); dt = time.perf_counter() - t0
    print(f"upload 3 x {n * 4 / 2**30:.1f} GiB columns: {dt * 1e3:8.1f} ms = {3 * n * 4 / dt / 1e9:6.2f} GB/s", flush=True)
    t.free()
ptr = eng.alloc(n * 4)
# device -> host: Engine.download() returns a numpy view of a pinned block the copy engine wrote (the first call of a size
# pins a fresh block; later calls re-use the context's cached blocks -- `a` is dropped before the next call)
for mib in (16, 64, 256, 1024):
    m = mib << 18
    for r in range(4):
        a = None
        t0 = time.perf_counter(); a = eng.download(ptr, m, np.float32); dt = time.perf_counter() - t0
        print(f"download {mib:5d} MiB ({'first call: pins a block' if r == 0 else 'cached pinned block'}): {dt * 1e3:8.2f} ms = {m * 4 / dt / 1e9:6.2f} GB/s", flush=True)
a = None
out = np.empty(n, dtype=np.float32)
for r in range(2):
    t0 = time.perf_counter(); eng._chk(eng.lib.hark_dev_download(eng.ctx, out.ctypes.data, ptr, out.nbytes)); dt = time.perf_counter() - t0
    print(f"download 1 GiB into PAGEABLE memory (hark_dev_download, the path of rounds 1-4): {dt * 1e3:8.1f} ms = {n * 4 / dt / 1e9:6.2f} GB/s", flush=True)

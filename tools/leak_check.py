"""Repeat every entry many times and watch device memory: the caching allocator must reach a steady state.
Usage: python tools/leak_check.py [reps]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from harkdb_amd.engine import Engine

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
eng = Engine(0)
rng = np.random.default_rng(0)
n = 3_000_000
t = eng.table_from_columns([rng.integers(0, 1 << 20, n).astype(np.uint32), rng.integers(0, 1 << 16, n).astype(np.uint32),
                            rng.integers(0, 2**32, n, dtype=np.uint64).astype(np.uint32), rng.random(n).astype(np.float32)])
small = eng.table_from_columns([rng.integers(0, 2**32, 200_000, dtype=np.uint64).astype(np.uint32), rng.integers(0, 9, 200_000).astype(np.uint32)])
wide = eng.table_from_columns([rng.integers(-2**62, 2**62, n).astype(np.int64), rng.integers(0, 9, n).astype(np.int32)])
wsmall = eng.table_from_columns([rng.integers(-2**62, 2**62, 100_000).astype(np.int64), rng.integers(0, 9, 100_000).astype(np.int32)])
ops = {
    "query_sel": lambda: eng.query_sel(t, [0, 2]),
    "filter_sel": lambda: eng.filter_sel(t, 3, ">", 0.5, [0, 1], want_row_index=True),
    "groupby dense": lambda: eng.query_groupby(t, 0, [1, 1], [2, 3]),
    "groupby hash": lambda: eng.query_groupby(t, 2, [1, 1], [2, 3]),
    "sort carried": lambda: eng.sort(t, 0, [0, 1]),
    "sort gathered": lambda: eng.sort(t, 3, [0, 1, 2], descending=True),
    "join u32 (partitioned)": lambda: eng.join(t, small, 2, 0, [0, 1], [1]),
    "join i64 (partitioned)": lambda: eng.join(wide, wsmall, 0, 0, [1], [1]),
    "filter_groupby": lambda: eng.filter_groupby(t, (3, ">", 0.5), 0, [("sum", 3), ("count", 0), ("max", 1)]),
    "filter_groupby_topk": lambda: eng.filter_groupby_topk(t, [(3, ">", 0.5)], 0, [("sum", 3), ("count", 0), ("max", 1)], [(2, ">", 1)], 1, True, 10),
    "topk": lambda: eng.topk(t, [(3, ">", 0.5)], 1, True, 10, [0, 1, 2]),
    "filter_groupby_subset": lambda: eng.filter_groupby_subset(t, [(3, ">", 0.5)], 0, np.arange(100, dtype=np.uint32), [("sum", 3), ("max", 1)]),
}
bad = False
for name, fn in ops.items():
    fn().free(); eng.sync()
    free0 = torch.cuda.mem_get_info(0)[0]
    for _ in range(reps):
        fn().free()
    eng.sync()
    free1 = torch.cuda.mem_get_info(0)[0]
    grew = (free0 - free1) / 2**20
    print(f"{name:22s} device memory change over {reps} repeats: {grew:8.1f} MiB", flush=True)
    bad = bad or grew > 64
sys.exit(1 if bad else 0)

"""ONE operator workload, exactly as bench.py's `configs.*` builds it, a few times -- the program to put behind
`rocprofv3 --kernel-trace` / `--pmc FETCH_SIZE` / `--pmc WRITE_SIZE` (tools/evidence.sh; summarise with tools/opmc.py).
Usage: python tools/op_one.py WORKLOAD [scale] [reps]
  join_c4    1/8 of BASELINE configs[3]: 1.25e8 probe rows x 1.25e7 unique i64 build keys, half of the probe rows match
  join_u32   the reference's join entry on u32 keys: 1e8 probe x 1e7 build rows, ~10 % match
  sort20     ORDER BY a 20-bit u32 key, key + one column out (1e8 rows)
  sort32     ORDER BY a 31-bit u32 key, key + one column out (1e8 rows)
  sort64     ORDER BY an i64 key spread over 63 bits, key + one column out (1e8 rows)
  sparse_gb  the headline statement over SPARSE i32 keys: 2^20 distinct keys k*odd spread over [-2^31, 2^31) (1e9 rows x scale)
  refgb      query_groupby(db, 0, [1, 1], [sum, max]) over 2^20 dense u32 keys (1e8 rows)
  refgb_hash query_groupby over 2^21 distinct u32 keys spread over [0, 2^32) (1e8 rows)
  c1 / c2    projection / WHERE + projection over 1e8 rows x 8 f32 columns
Every repetition prints its wall time; the LAST repetition is the one tools/opmc.py reads."""
import os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from harkdb_amd.engine import Engine
import bench

what = sys.argv[1]
scale = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
dev = torch.device("cuda", 0)
eng = Engine(0)
w = bench.WORKLOADS[what](torch, eng, dev, scale)
mark = eng.alloc(64)
for r in range(reps):
    # a marker dispatch in front of every repetition (gen_columns_kernel over one row: no statement launches it): tools/opmc.py
    # takes the dispatches behind the LAST marker as one repetition, whatever the statement's kernel sequence looks like
    eng.gen_columns(1, 0, 1, 1, True, None, mark, None)
    eng.sync(); t0 = time.perf_counter(); res = w["run"](); eng.sync()
    ms = (time.perf_counter() - t0) * 1e3
    print(f"{what}: {ms:.3f} ms  out={getattr(res, 'shape', None)}  alg {w['bytes'](res) / ms / 1e9:.3f} TB/s", flush=True)
    if hasattr(res, "free"):
        res.free()

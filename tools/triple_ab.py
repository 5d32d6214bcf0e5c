"""A/B of the passes that serve SUM(a), MAX(b), MIN(c), COUNT(*) ... WHERE p > t GROUP BY k for every group (BASELINE
configs[4] share: 5e8 rows, 2^20 groups): one triple pass / pair + single (HARK_NO_TRIPLE_PASS) / three single passes
(HARK_NO_PAIR_PASS), interleaved in one process, at three selectivities.  Results are compared bit for bit.
Usage: python tools/triple_ab.py [scale] [reps]"""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from harkdb_amd.engine import Engine
from harkdb_amd import dist as hd
import bench

scale = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
dev = torch.device("cuda", 0)
eng = Engine(0)
stream = hd.share_stream(eng, dev)
n, G = int(5e8 * scale) // 4 * 4, 1 << 20
cols = [torch.empty(n, dtype=torch.float32, device=dev) for _ in range(4)]
key = torch.empty(n, dtype=torch.int32, device=dev)
eng.gen_columns(bench.SEED, 0, n, G, False, cols[0].data_ptr(), key.data_ptr(), cols[1].data_ptr())
eng.gen_columns(bench.SEED + 2, 0, n, G, False, cols[2].data_ptr(), None, cols[3].data_ptr())
t = eng.table_from_device(n, [key.data_ptr()] + [c.data_ptr() for c in cols], [np.int32] + [np.float32] * 4)
modes = [("triple pass", {}), ("pair + single", {"HARK_NO_TRIPLE_PASS": "1"}), ("three single passes", {"HARK_NO_PAIR_PASS": "1"})]
for thr in (0.5, 0.1, 0.9):
    ref = None
    for name, env in modes:
        os.environ.update(env)
        def run():
            eng.filter_groupby(t, [(1, ">", thr)], 0, [("sum", 2), ("max", 3), ("min", 4), ("count", 0)]).free()

        ms = bench.event_ms(torch, run, warm=1, reps=reps)
        r = eng.filter_groupby(t, [(1, ">", thr)], 0, [("sum", 2), ("max", 3), ("min", 4), ("count", 0)])
        passes, got = eng.last_groupby_passes(), [c.copy() for c in r.columns()]
        r.free()
        for k in env:
            del os.environ[k]
        same = True if ref is None else all(np.array_equal(x, y) for x, y in zip(ref, got))
        if ref is None:
            ref = got
        print(f"where p > {thr}: {name:22s} {ms:7.3f} ms  passes {passes}  rows {len(got[0])}  equal to the first mode: {same}", flush=True)

# ---- the reference's entry: query_groupby(db, 0, [1, 2, 3], [sum, max, min]) over 1e8 rows x 4 u32 columns, 2^20 dense keys:
#      every row survives (no WHERE in the reference), 14 B per row of partition traffic both ways
t.free()
del cols, key
torch.cuda.empty_cache()
n = int(1e8 * scale) // 4 * 4
u = [torch.randint(0, 2**31, (n,), dtype=torch.int32, device=dev) for _ in range(3)]
kk = torch.randint(0, G, (n,), dtype=torch.int32, device=dev)
t = eng.table_from_device(n, [kk.data_ptr()] + [c.data_ptr() for c in u], [np.uint32] * 4)
ref = None
for name, env in modes:
    os.environ.update(env)

    def run():
        eng.query_groupby(t, 0, [1, 2, 3], [2, 3, 4]).free()

    ms = bench.event_ms(torch, run, warm=1, reps=reps)
    got = eng.query_groupby(t, 0, [1, 2, 3], [2, 3, 4]).to_numpy(np.uint32)
    for k in env:
        del os.environ[k]
    same = True if ref is None else np.array_equal(ref, got)
    if ref is None:
        ref = got
    print(f"query_groupby sum, max, min of three columns, {n} rows: {name:22s} {ms:7.3f} ms  rows {len(got)}  equal to the first mode: {same}", flush=True)

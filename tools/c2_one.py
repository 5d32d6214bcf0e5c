"""BASELINE configs[1] as bench.py's C2_filter_proj builds it: SELECT rowid, c0, c2 FROM t8 WHERE c1 > 0.5 over 1e8 rows x 8 f32
columns, HIP-event timed (3 warm-ups, median of 10).  Usage: python tools/c2_one.py [rows]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from harkdb_amd.engine import Engine
from harkdb_amd import dist as hd
n2 = int(float(sys.argv[1])) if len(sys.argv) > 1 else 100_000_000
dev = torch.device("cuda", 0)
eng = Engine(0)
hd.share_stream(eng, dev)
cols = [torch.empty(n2, dtype=torch.float32, device=dev) for _ in range(8)]
for j in range(0, 8, 2):
    eng.gen_columns(0x4861726B4442 + j, 0, n2, 1 << 20, False, cols[j].data_ptr(), None, cols[j + 1].data_ptr())
t8 = eng.table_from_device(n2, [c.data_ptr() for c in cols], [np.float32] * 8, keepalive=cols)
shape = [None]
def c2():
    r = eng.filter_sel(t8, 1, ">", 0.5, [0, 2], want_row_index=True); shape[0] = r.shape; r.free()
for thr, name in ((0.5, "sel 0.5"),):
    for _ in range(3): c2()
    torch.cuda.synchronize(); ts = []
    for _ in range(10):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); c2(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    ts.sort(); ms = ts[5]; surv = shape[0][0]
    print(f"C2 {name}: {ms:.4f} ms  survivors {surv}  {(12.0 * n2 + 16.0 * surv) / ms / 1e6 / 8000:.3f} of peak", flush=True)

"""One GPU's share of BASELINE configs[3] exactly as bench.py's C4_join_share builds it (1.25e8 probe rows, 1.25e7 unique
i64 build keys, half of the probe rows match), a few times (for rocprofv3).  Usage: python tools/join_c4.py [scale]"""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from harkdb_amd.engine import Engine
scale = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
dev = torch.device("cuda", 0)
eng = Engine(0)
SEED = 0x4861726B4442
n4, s4 = int(1.25e8 * scale), int(1.25e7 * scale)
mul = int(os.environ.get("C4_MUL", "-7046029254386353131"))
bk = torch.arange(s4, dtype=torch.int64, device=dev) * mul
j = torch.empty((n4 + 3) // 4 * 4, dtype=torch.int32, device=dev)
eng.gen_columns(SEED + 21, 0, n4, 2 * s4, True, None, j.data_ptr(), None)
j = j[:n4]
pk = j.to(torch.int64) * mul
prow, brow = torch.arange(n4, dtype=torch.int32, device=dev), torch.arange(s4, dtype=torch.int32, device=dev)
hits = int((j < s4).sum().item())
tp = eng.table_from_device(n4, [pk.data_ptr(), prow.data_ptr()], [np.int64, np.int32], keepalive=(pk, prow))
tb = eng.table_from_device(s4, [bk.data_ptr(), brow.data_ptr()], [np.int64, np.int32], keepalive=(bk, brow))
torch.cuda.synchronize()
for r in range(4):
    eng.sync(); t0 = time.perf_counter(); res = eng.join(tp, tb, 0, 0, [1], [1]); eng.sync()
    print(f"C4 join {n4} x {s4}: {(time.perf_counter() - t0) * 1e3:.3f} ms out={res.shape} expected pairs {hits}", flush=True)
    res.free()

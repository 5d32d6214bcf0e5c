"""A/B of the projection kernel's memory instructions on ONE box in ONE process (C1 went 0.729 -> 0.665 of
peak between two driver runs -- the card, or the non-temporal stores?).  SELECT c0, c2 over 1e8 rows x 8 f32 columns
(bench.py's C1_projection), every variant of copy_columns_kernel (HARK_COPY_VARIANT = <nt loads><nt stores><loads in
flight>), HIP-event timed, interleaved over several rounds so that drift hits every arm alike; C2 (WHERE + projection)
beside it for the record.  Usage: python tools/copy_ab.py [rows] [rounds]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from harkdb_amd.engine import Engine
from harkdb_amd import dist as hd

scale = (float(sys.argv[1]) if len(sys.argv) > 1 else 1e8) / 1e8
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 4
dev = torch.device("cuda", 0)
eng = Engine(0)
hd.share_stream(eng, dev)
w = bench.w_c2(torch, eng, dev, scale)
t8, n2 = w["keep"][0], w["rows"]
print(f"device: {torch.cuda.get_device_properties(0).name}  pci {os.popen('rocm-smi --showbus 2>/dev/null | grep -m1 GPU').read().strip()}", flush=True)
variants = ["112", "102", "012", "002", "114", "104", "014", "004"]
res = {v: [] for v in variants}
c2 = []
for r in range(rounds):
    for v in variants:
        os.environ["HARK_COPY_VARIANT"] = v
        res[v].append(bench.event_ms(torch, lambda: eng.query_sel(t8, [0, 2]).free(), warm=2, reps=7))
    os.environ.pop("HARK_COPY_VARIANT")
    c2.append(bench.event_ms(torch, lambda: w["run"]().free(), warm=2, reps=7))
for v in variants:
    ms = sorted(res[v])
    print(f"C1 variant nt-loads={v[0]} nt-stores={v[1]} in-flight={v[2]}: median {ms[len(ms) // 2]:.4f} ms  min {ms[0]:.4f}  max {ms[-1]:.4f}"
          f"  {16.0 * n2 / ms[len(ms) // 2] / 1e6 / 8000:.3f} of peak" + ("   <- default" if v == "112" else ""), flush=True)
# what the runtime's own device-to-device copy reaches on the same two columns (hipMemcpyDtoDAsync through torch), same process
cols = w["keep"][1]
dsts = [torch.empty_like(cols[0]), torch.empty_like(cols[2])]
rt = sorted(bench.event_ms(torch, lambda: (dsts[0].copy_(cols[0]), dsts[1].copy_(cols[2])), warm=2, reps=7) for _ in range(rounds))
print(f"runtime copy of the same two columns (torch .copy_ = hipMemcpyDtoDAsync x 2): median {rt[len(rt) // 2]:.4f} ms  "
      f"{16.0 * n2 / rt[len(rt) // 2] / 1e6 / 8000:.3f} of peak", flush=True)
ms = sorted(c2)
surv = w["run"]()
print(f"C2 WHERE + projection: median {ms[len(ms) // 2]:.4f} ms  min {ms[0]:.4f}  max {ms[-1]:.4f}  "
      f"{w['bytes'](surv) / ms[len(ms) // 2] / 1e6 / 8000:.3f} of peak", flush=True)

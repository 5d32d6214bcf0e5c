"""A/B of the radix passes' ranking inside ONE process: one returning LDS atomic per key (ARANK, the default where the
device hands them out in lane order) against the lane-mask exchange (HARK_SORT_MASKRANK=1), interleaved rounds, HIP-event
medians.  Usage: python tools/sort_ab.py [rounds]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from harkdb_amd.engine import Engine
from harkdb_amd import dist as hd

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
dev = torch.device("cuda", 0)
eng = Engine(0)
hd.share_stream(eng, dev)
print(f"pci {os.popen('rocm-smi --showbus 2>/dev/null | grep -m1 GPU').read().strip()}", flush=True)
for name in ("sort20", "sort32", "sort64", "join_u32", "refgb_hash"):
    w = bench.WORKLOADS[name](torch, eng, dev, 1.0)
    res = {"0": [], "1": []}
    for r in range(rounds):
        for knob in ("0", "1"):
            os.environ["HARK_SORT_MASKRANK"] = knob
            res[knob].append(bench.event_ms(torch, lambda: w["run"]().free(), warm=2, reps=7))
    os.environ.pop("HARK_SORT_MASKRANK")
    a, m = sorted(res["0"]), sorted(res["1"])
    print(f"{name:12s} atomic rank {a[len(a) // 2]:.4f} ms (min {a[0]:.4f})   mask exchange {m[len(m) // 2]:.4f} ms (min {m[0]:.4f})   ratio {a[len(a) // 2] / m[len(m) // 2]:.3f}", flush=True)
    del w
    torch.cuda.empty_cache()
print("lds_lane_order (device check):", eng.lib.hark_context_lds_lane_order(eng.ctx))

"""configs[2] as written (no filter: p = NULL) through the producer's existing geometries, interleaved in one process:
pairfmt 0 (256 buckets, rings of 96 six-byte pairs) against pairfmt 3 (128 buckets, rings of 144 one-word entries: up to two
units leave per sweep), each with the sweep period left to the kernel and forced.   python tools/nofilter_knobs.py [rows]"""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from harkdb_amd.engine import Engine, FgbPlan
from harkdb_amd import dist as hd
import bench

N = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10**9
G = 1 << 20
dev = torch.device("cuda", 0)
eng = Engine(0)
hd.share_stream(eng, dev)
p, k, v = (torch.empty(N, dtype=dt, device=dev) for dt in (torch.float32, torch.int32, torch.float32))
eng.gen_columns(bench.SEED, 0, N, G, True, p.data_ptr(), k.data_ptr(), v.data_ptr())
so, co = torch.empty(G, dtype=torch.float32, device=dev), torch.empty(G, dtype=torch.int64, device=dev)
VARIANTS = [dict(), dict(pairfmt=3), dict(pairfmt=3, period=2), dict(pairfmt=3, period=3), dict(pairfmt=3, period=1), dict(period=1), dict(pairfmt=1)]
for rnd in range(2):
    for kn in VARIANTS:
        for name, pp, cmp, thr in (("p = NULL", None, ">", 0.5), ("p > 0.5 ", p.data_ptr(), ">", 0.5)):
            plan = FgbPlan(eng, N, G, timing=1, **kn)

            def step():
                plan.reset()
                plan.run(pp, cmp, thr, k.data_ptr(), v.data_ptr(), N)
                plan.finish(so.data_ptr(), co.data_ptr())

            ms = bench.event_ms(torch, step, warm=2, reps=5)
            kms, kl = plan.timing()
            print(f"round {rnd} {str(kn):32s} {name}: step {ms:.3f} ms  producer {kms['producer'] / max(1, kl['producer']):.3f}  consumer {kms['consumer'] / max(1, kl['consumer']):.3f}", flush=True)
            plan.free()

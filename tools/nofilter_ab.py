"""configs[2] as written (no filter: p = NULL at the C ABI) against the same rows through the predicate path with a
predicate every row passes (p >= 0), interleaved in one process: is the kNoPred instantiation of the producer slower than
the predicate one (bench line r05a: 3.84 against 3.28 ms producer)?   python tools/nofilter_ab.py [rows] [period]"""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from harkdb_amd.engine import Engine, FgbPlan
from harkdb_amd import dist as hd
import bench

N = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10**9
period = int(sys.argv[2]) if len(sys.argv) > 2 else 0
G = 1 << 20
dev = torch.device("cuda", 0)
eng = Engine(0)
hd.share_stream(eng, dev)
p, k, v = (torch.empty(N, dtype=dt, device=dev) for dt in (torch.float32, torch.int32, torch.float32))
eng.gen_columns(bench.SEED, 0, N, G, True, p.data_ptr(), k.data_ptr(), v.data_ptr())
so, co = torch.empty(G, dtype=torch.float32, device=dev), torch.empty(G, dtype=torch.int64, device=dev)
for rnd in range(3):
    for name, pp, cmp in (("p = NULL", None, ">"), ("p >= 0  ", p.data_ptr(), ">="), ("k != NaN", k.data_ptr(), "!="), ("v != NaN", v.data_ptr(), "!="), ("p > 0.5 ", p.data_ptr(), ">")):
        kn = {"timing": 1}
        if period:
            kn["period"] = period
        plan = FgbPlan(eng, N, G, **kn)

        def step():
            plan.reset()
            plan.run(pp, cmp, float("nan") if cmp == "!=" else 0.0 if cmp == ">=" else 0.5, k.data_ptr(), v.data_ptr(), N)   # x != NaN holds for every x: the key / value column as an always-true "predicate"
            plan.finish(so.data_ptr(), co.data_ptr())

        ms = bench.event_ms(torch, step, warm=2, reps=7)
        kms, kl = plan.timing()
        print(f"round {rnd} {name}: step {ms:.3f} ms  producer {kms['producer'] / max(1, kl['producer']):.3f}  consumer {kms['consumer'] / max(1, kl['consumer']):.3f}  rows out {int(co.sum().item())}", flush=True)
        plan.free()

"""A/B timing of fused filter->group-by plan configurations (grid, shift, period, pairfmt, chunk_rows, slack_pct, G) with the library's own HIP-event timers.
Interleaved rounds (config order repeated R times) so clock drift hits every arm alike.
Usage: python tools/fgb_ablate.py [N] ["k=v,k=v;k=v,..."]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from harkdb_amd.engine import Engine, FgbPlan

N = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1 << 28
G = int(sys.argv[3]) if len(sys.argv) > 3 else 1 << 20
spec = sys.argv[2] if len(sys.argv) > 2 else "grid=512;grid=1024"
configs = [dict((kv.split("=")[0], int(kv.split("=")[1])) for kv in c.split(",") if kv) for c in spec.split(";")]
R = 8
eng = Engine(0)
p, k, v = eng.alloc(N * 4), eng.alloc(N * 4), eng.alloc(N * 4)
eng.gen_columns(0x4861726B4442, 0, N, G, True, p, k, v)
plans = []
shared = {}
for c in configs:
    c.pop("tag", None)                    # tag=<n>: a second, separate plan of an otherwise identical configuration
    key = tuple(sorted(c.items())) + ((("tag", len(plans)),) if "tag=" in spec.split(";")[len(plans)] else ())
    if key not in shared:
        cc = dict(c)
        g = cc.pop("G", G)
        pl = FgbPlan(eng, N, g, algo=cc.pop("algo", 3), chunk_rows=cc.pop("chunk_rows", N))
        for kk, vv in cc.items():
            pl.set(kk, vv)
        pl.set("timing", 1)
        shared[key] = pl
    plans.append(shared[key])
res = [[] for _ in plans]
ref = None            # every configuration must produce the same table
so, co = eng.alloc(G * 4), eng.alloc(G * 8)
for i, pl in enumerate(plans):
    if pl.G != G:
        continue
    pl.reset(); pl.run(p, ">", 0.5, k, v, N); pl.finish(so, co)
    got = (eng.download(so, G, np.float32), eng.download(co, G, np.int64))
    if ref is None:
        ref = got
        print("reference table: %d survivors" % int(got[1].sum()), flush=True)
    elif not (np.array_equal(ref[0], got[0]) and np.array_equal(ref[1], got[1])):
        print("MISMATCH: configuration %d (%s) differs from configuration 0" % (i, spec.split(";")[i]), flush=True)
    pl.timing()
for r in range(R + 1):
    for i, pl in enumerate(plans):
        pl.reset(); pl.run(p, ">", 0.5, k, v, N); ms, cnt = pl.timing()
        if r:
            res[i].append((ms["producer"] + ms["single"], ms["consumer"]))
for c, rr in zip(spec.split(";"), res):
    a = np.array(rr)
    tot = a.sum(axis=1)
    print(f"{c:40s} producer min {a[:,0].min():7.3f} med {np.median(a[:,0]):7.3f} | consumer min {a[:,1].min():7.3f} med {np.median(a[:,1]):7.3f} | "
          f"total min {tot.min():7.3f} med {np.median(tot):7.3f} ms  -> {(12*N+16*G)/np.median(tot)/1e9/8:.3f} of 8 TB/s", flush=True)

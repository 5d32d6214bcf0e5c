"""Randomized differential stress of the partition path (producer rings, sweeps, slab overflow, chunking) against numpy.
Integer-valued f32 data: sums and counts must match exactly.  Usage: python tools/fgb_stress.py [seconds] [seed]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from harkdb_amd.engine import Engine, FgbPlan

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
eng = Engine(0)
t_end, cases, worst = time.time() + budget, 0, None
t_note = time.time() + 60
while time.time() < t_end:
    if time.time() > t_note:                                  # a sign of life per minute (a silent GPU job is taken to be hung)
        print(f"... {cases} cases so far", flush=True); t_note = time.time() + 60
    G = int(rng.choice([8193, 10_000, 65_536, 300_001, 1 << 20, (1 << 21) - 5]))
    n = int(rng.integers(1, 3_000_000))
    dist = rng.choice(["uniform", "zipf", "one bucket", "few keys", "sorted"])
    if dist == "uniform":
        k = rng.integers(0, G, size=n)
    elif dist == "zipf":
        k = np.minimum(rng.zipf(1.3, size=n) - 1, G - 1)
    elif dist == "one bucket":
        base = int(rng.integers(0, max(1, G - 4096)))
        k = base + rng.integers(0, min(4096, G - base), size=n)
    elif dist == "few keys":
        k = rng.choice(rng.integers(0, G, size=int(rng.integers(1, 40))), size=n)
    else:
        k = np.sort(rng.integers(0, G, size=n))
    k = k.astype(np.int32)
    v = rng.integers(0, 16, size=n).astype(np.float32)
    p = rng.random(n).astype(np.float32)
    thr = float(rng.choice([-1.0, 0.02, 0.5, 0.98, 2.0]))
    use_pred = bool(rng.random() < 0.8)
    count_only = bool(rng.random() < 0.25)                    # no value column: keys-only partition format
    knobs = {"algo": int(rng.choice([3, 3, 3, 2])) if count_only else 3}
    if rng.random() < 0.5:
        knobs["chunk_rows"] = int(rng.choice([8192, 1 << 16, 1 << 20]))
    if rng.random() < 0.3:
        knobs["slack_pct"] = int(rng.choice([1, 20, 100]))
    if rng.random() < 0.3:
        knobs["pairfmt"] = int(rng.choice([1, 2]))
    if rng.random() < 0.2:
        knobs["grid"] = int(rng.choice([1, 7, 64, 256]))
    plan = FgbPlan(eng, n, G, **knobs)
    dp, dk, dv = eng.alloc(max(n * 4, 16)), eng.alloc(max(n * 4, 16)), eng.alloc(max(n * 4, 16))
    eng.upload(dp, p); eng.upload(dk, k); eng.upload(dv, v)
    plan.reset(); plan.run(dp if use_pred else None, ">", thr, dk, None if count_only else dv, n)
    ds, dc = eng.alloc(G * 4), eng.alloc(G * 8)
    plan.finish(ds, dc)
    gs, gc = eng.download(ds, G, np.float32), eng.download(dc, G, np.int64)
    eng.free(ds); eng.free(dc)
    keep = (p > thr) if use_pred else np.ones(n, bool)
    ec = np.bincount(k[keep], minlength=G).astype(np.int64)
    es = np.bincount(k[keep], weights=v[keep].astype(np.float64), minlength=G).astype(np.float32)
    if count_only:
        es = np.zeros(G, dtype=np.float32)
    ok = np.array_equal(gc, ec) and np.array_equal(gs, es)
    cases += 1
    if not ok:
        print("MISMATCH", dict(G=G, n=n, dist=dist, thr=thr, use_pred=use_pred, count_only=count_only, **knobs), flush=True)
        worst = True
        break
    plan.free(); eng.free(dp); eng.free(dk); eng.free(dv)
print(f"{cases} random cases, {'FAILED' if worst else 'all exact'}", flush=True)
sys.exit(1 if worst else 0)

"""Random SQL statements through FutharkContext.sql_columns() against pandas on the same frame.
Usage: python tools/sql_stress.py [seconds] [seed]"""
import os, sys, time
import numpy as np, pandas as pd
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from harkdb_amd import FutharkContext

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
fc = FutharkContext(sql_mode=True)
OPS = {">": "gt", ">=": "ge", "<": "lt", "<=": "le", "=": "eq", "!=": "ne"}
cases, bad, t_end = 0, None, time.time() + budget
t_note = time.time() + 60
while time.time() < t_end and bad is None:
    if time.time() > t_note:                                  # a sign of life per minute (a silent GPU job is taken to be hung)
        print(f"... {cases} statements so far", flush=True); t_note = time.time() + 60
    n = int(rng.choice([7, 5000, 300_000, 1_200_000]))
    df = pd.DataFrame({
        "a": rng.integers(-4, 5, n).astype(np.int32), "b": rng.integers(0, int(rng.choice([3, 50, 4000])), n).astype(np.int32),
        "s": (rng.integers(0, int(rng.choice([10, 3000, 200_000])), n) * 1_000_003 % (2**31)).astype(np.int32),
        "d": rng.integers(0, int(rng.choice([20_000, 300_000])), n).astype(np.int32),       # dense keys beyond the LDS path
        "x": rng.integers(0, 64, n).astype(np.float32), "y": rng.integers(-1000, 1000, n).astype(np.int32), "p": rng.random(n).astype(np.float32),
        "z": rng.integers(0, 2**31, n).astype(np.int32), "w": (rng.integers(-300, 300, n) / 2).astype(np.float32)})
    fc.create_table("t", df)
    for _ in range(12):
        where_sql, mask = "", np.ones(n, bool)

        def leaf():
            col = str(rng.choice(["p", "y", "a"])); op = str(rng.choice(list(OPS)))
            val = float(np.float32(rng.random())) if col == "p" else int(rng.integers(-3, 4)) if col == "a" else int(rng.integers(-900, 900))
            return f"{col} {op} {val!r}", getattr(df[col], OPS[op])(np.float32(val) if col == "p" else val).to_numpy()

        def tree(depth):
            """OR / NOT / IN / parentheses / BETWEEN / two columns (round 6: what the reference's parser accepts, parse.py:27)."""
            r = rng.random()
            if depth == 0 or r < 0.35:
                return leaf()
            if r < 0.45:
                col = str(rng.choice(["a", "b", "y"])); vals = [int(v) for v in rng.integers(-4, 60, size=int(rng.integers(1, 5)))]
                neg = rng.random() < 0.3
                m = df[col].isin(vals).to_numpy()
                return f"{col} {'not ' if neg else ''}in ({', '.join(map(str, vals))})", (~m if neg else m)
            if r < 0.52:
                op = str(rng.choice(list(OPS)))
                return f"a {op} b", getattr(df["a"], OPS[op])(df["b"]).to_numpy()
            if r < 0.6:
                lo, hi = sorted(int(v) for v in rng.integers(-900, 900, size=2)); neg = rng.random() < 0.3
                m = ((df.y >= lo) & (df.y <= hi)).to_numpy()
                return f"y {'not ' if neg else ''}between {lo} and {hi}", (~m if neg else m)
            if r < 0.7:
                t_, m = tree(depth - 1)
                return f"not ({t_})", ~m
            kids = [tree(depth - 1) for _ in range(int(rng.integers(2, 4)))]
            if r < 0.85:
                return "(" + " and ".join(f"({t_})" for t_, _ in kids) + ")", np.logical_and.reduce([m for _, m in kids])
            return "(" + " or ".join(f"({t_})" for t_, _ in kids) + ")", np.logical_or.reduce([m for _, m in kids])

        if rng.random() < 0.7:
            text, mask = leaf() if rng.random() < 0.4 else tree(3)
            where_sql = f" where {text}"
        sub = df[mask]
        kind = rng.choice(["group", "multi", "distinct", "select"])
        try:
            if kind in ("group", "multi"):
                keys = [str(rng.choice(["a", "b", "s", "d", "d"]))] if kind == "group" else [str(c) for c in rng.choice(["a", "b", "s"], size=2, replace=False)]
                # (aggregates of three and more different columns: triple and pair passes on the dense path)
                aggs = [("sum", "x"), ("count", "*"), ("avg", "x"), ("max", "y"), ("min", "y"), ("sum", "y"), ("max", "x"), ("min", "z"), ("max", "z"), ("max", "w"), ("min", "w"), ("sum", "w")]
                pick = [aggs[i] for i in rng.choice(len(aggs), size=int(rng.integers(1, 7)), replace=False)]
                sel = keys + [f"{f}({c})" for f, c in pick]
                stmt = f"select {', '.join(sel)} from t{where_sql} group by {', '.join(keys)}"
                g = sub.groupby(keys, sort=True)
                exp = g.size().reset_index(name="__n")[keys]
                for f, c in pick:
                    if f == "count":
                        exp[f"{f}({c})"] = g.size().to_numpy()
                    else:
                        exp[f"{f}({c})"] = getattr(g[c], {"sum": "sum", "avg": "mean", "max": "max", "min": "min"}[f])().to_numpy()
                if rng.random() < 0.3 and len(exp):
                    thr = int(np.median(g.size().to_numpy()))
                    stmt += f" having count(*) >= {thr}"
                    exp = exp[g.size().to_numpy() >= thr]
                if kind == "group" and rng.random() < 0.4:
                    # ORDER BY an aggregate (exact ones: integer-valued data) or the key; ties keep key order (stable)
                    cand = [keys[0]] + [f"{f}({c})" for f, c in pick if f != "avg"]
                    oc = str(rng.choice(cand)); desc = bool(rng.random() < 0.5)
                    stmt += f" order by {oc}{' desc' if desc else ''}"
                    exp = exp.sort_values(oc, ascending=not desc, kind="stable")
            elif kind == "distinct":
                keys = [str(c) for c in rng.choice(["a", "b", "s"], size=int(rng.integers(1, 3)), replace=False)]
                stmt = f"select distinct {', '.join(keys)} from t{where_sql}"
                exp = sub[keys].drop_duplicates().sort_values(keys)
            else:
                cols = [str(c) for c in rng.choice(["a", "b", "y", "x"], size=2, replace=False)]
                stmt = f"select {', '.join(cols)} from t{where_sql}"
                exp = sub[cols]
                if rng.random() < 0.6:
                    ok = [c for c in cols if c != "x"]
                    desc = bool(rng.random() < 0.5)
                    stmt += " order by " + ", ".join(f"{c}{' desc' if desc else ''}" for c in ok)
                    exp = exp.sort_values(ok, ascending=not desc, kind="stable")
            lim = int(rng.integers(1, 50)) if rng.random() < 0.3 else None
            if lim is not None:
                stmt += f" limit {lim}"
                exp = exp.head(lim)
            names, cols_out = fc.sql_columns(stmt)
            ok_all = names == list(exp.columns) or kind == "select" or kind == "distinct"
            for got, name in zip(cols_out, exp.columns):
                e = exp[name].to_numpy()
                if len(got) != len(e):
                    ok_all = False
                elif got.dtype.kind == "f" or e.dtype.kind == "f":
                    ok_all = ok_all and np.allclose(got.astype(np.float64), e.astype(np.float64), rtol=2e-6, atol=1e-6)
                else:
                    ok_all = ok_all and np.array_equal(got.astype(np.int64), e.astype(np.int64))
            # ... and the reference's return shape: sql() builds ONE matrix on the device wherever the result is a device result as it
            # stands (round 5); element for element and in dtype it must be the typed columns interleaved by numpy
            if ok_all and cols_out and rng.random() < 0.5:
                dts = {c.dtype for c in cols_out}
                dt = dts.pop() if len(dts) == 1 else np.result_type(*[c.dtype for c in cols_out])
                want = np.empty((len(cols_out[0]), len(cols_out)), dtype=dt)
                for j, c in enumerate(cols_out):
                    want[:, j] = c
                m = fc.sql(stmt)
                ok_all = m.dtype == want.dtype and m.shape == want.shape and np.array_equal(m, want, equal_nan=True)
            cases += 1
            if not ok_all:
                bad = (n, stmt)
                break
        except Exception as ex:                                  # noqa: BLE001
            bad = (n, stmt, repr(ex)[:200])
            break
    fc.drop_table("t")
print(f"{cases} random statements: {'MISMATCH ' + str(bad) if bad else 'all match pandas'}", flush=True)
sys.exit(1 if bad else 0)

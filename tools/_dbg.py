import os, sys, numpy as np, pandas as pd
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from harkdb_amd import FutharkContext
fc = FutharkContext(sql_mode=True)
rng = np.random.default_rng(3)
n = 1_200_000
for K in (10, 3000, 200_000):
    for B in (3, 50, 4000):
        df = pd.DataFrame({"b": rng.integers(0, B, n).astype(np.int32), "s": (rng.integers(0, K, n) * 1_000_003 % (2**31)).astype(np.int32),
                           "y": rng.integers(-1000, 1000, n).astype(np.int32), "z": rng.integers(0, 2**31, n).astype(np.int32)})
        fc.create_table("t", df)
        for env in ("", "1"):
            if env: os.environ["HARK_SORT_NO_MSD"] = "1"
            else: os.environ.pop("HARK_SORT_NO_MSD", None)
            names, cols = fc.sql_columns("select s, b, min(z), max(y) from t group by s, b")
            g = df.groupby(["s", "b"], sort=True)
            exp = g.size().reset_index(name="n")
            ok = len(cols[0]) == len(exp) and np.array_equal(cols[0], exp["s"].to_numpy()) and np.array_equal(cols[1], exp["b"].to_numpy()) \
                and np.array_equal(cols[2], g["z"].min().to_numpy()) and np.array_equal(cols[3], g["y"].max().to_numpy())
            print(K, B, "nomsd" if env else "msd", "groups", len(exp), "got", len(cols[0]), "OK" if ok else "MISMATCH", flush=True)

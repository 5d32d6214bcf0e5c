"""Latency of the reference's own statements on its 7-row data.csv (BASELINE configs[0]).  Usage: python tools/small_latency.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from harkdb_amd import FutharkContext
fc = FutharkContext()
fc.create_table("game_1", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "data.csv"))
for stmt in ("select col1, col3 from game_1", "select col1,  max(col3) from game_1 group by col1",
             "select col1, col3 from game_1 where col2 > 3", "select col1, sum(col3), count(*) from game_1 group by col1 order by col1 desc limit 2"):
    fc.sql(stmt)
    ts = []
    for _ in range(200):
        t0 = time.perf_counter(); r = fc.sql(stmt); ts.append(time.perf_counter() - t0)
    ts.sort()
    print(f"{ts[len(ts)//2]*1e6:8.1f} us median  {ts[0]*1e6:8.1f} us min   {stmt}   -> {r.tolist()[:3]}")

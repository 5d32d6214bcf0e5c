"""Per-phase cycle counts of the join's order kernel (debug build libhark_jprof.so: tools/ab_build.sh jprof -DHARK_JORDER_PROF k_hjoin.hip)."""
import ctypes as C, os, sys
os.environ["HARK_LIB"] = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "harkdb_amd", "libhark_jprof.so")
sys.argv = [sys.argv[0]]
exec(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "join_c4.py")).read().replace("for r in range(4):", "for r in range(2):"))
lib = C.CDLL(os.environ["HARK_LIB"])
out = (C.c_ulonglong * 16)()
assert lib.hark_debug_jprof(out) == 0
names = ["coarse scan", "sub-round table", "binning", "zero fine", "count sweep", "fine scan", "placement sweep", "tie place + write-out"]
tot = sum(out[:8])
for n, v in zip(names, out[:8]):
    print(f"{n:24s} {v / 512 / 2 / 100e6 * 1e6 * 1e3:9.1f} us per bucket (at 100 MHz counter)  {100.0 * v / tot:5.1f} %")

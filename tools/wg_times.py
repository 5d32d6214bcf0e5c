"""Per-workgroup start / end times of the producer of ONE headline step (needs a debug build of libhark.so that exports
hark_debug_wgtimes: HARK_LIB=...).  Usage: HARK_LIB=$PWD/harkdb_amd/libhark_dbg.so python tools/wg_times.py"""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from harkdb_amd.engine import Engine, FgbPlan
from harkdb_amd import _ffi
N, G = 1_000_000_000, 1 << 20
eng = Engine(0)
lib = _ffi.load()
p, k, v = eng.alloc(N * 4), eng.alloc(N * 4), eng.alloc(N * 4)
eng.gen_columns(0x4861726B4442, 0, N, G, False, p, k, v)
plan = FgbPlan(eng, N, G, timing=1)
sums, counts = eng.alloc(G * 4), eng.alloc(G * 8)
buf = (ctypes.c_ulonglong * (3 * 1024))()
for rep in range(4):
    plan.reset(); plan.run(p, ">", 0.5, k, v, N); plan.finish(sums, counts)
    ms, cnt = plan.timing()
    lib.hark_debug_wgtimes(buf)
    a = np.array(buf[:], dtype=np.int64).reshape(3, 1024)[:, :256]
    t0, t1, xcc = a[0], a[1], a[2] & 0xF
    base = t0.min()
    dur = (t1 - t0) / 100.0          # wall_clock64 ticks at 100 MHz -> microseconds
    end = (t1 - base) / 100.0
    start = (t0 - base) / 100.0
    print(f"rep {rep}: producer {ms['producer']:.3f} ms; workgroup durations (us): min {dur.min():.0f} mean {dur.mean():.0f} max {dur.max():.0f}; "
          f"start skew max {start.max():.0f} us; end: first {end.min():.0f} last {end.max():.0f}")
    if rep == 3:
        for x in range(8):
            m = xcc == x
            if m.any():
                print(f"   XCC {x}: {m.sum():3d} workgroups, duration mean {dur[m].mean():.0f} max {dur[m].max():.0f} us, end mean {end[m].mean():.0f} max {end[m].max():.0f}")
        srt = np.sort(end)
        print("   end-time percentiles (us): " + ", ".join(f"p{q}={np.percentile(end, q):.0f}" for q in (0, 10, 50, 90, 99, 100)))

"""Per-kernel summary of a rocprofv3 --kernel-trace csv directory: python tools/ktrace.py DIR [last_fraction]
Counts calls and total / average time per kernel name over the LAST fraction of the dispatches (default 0.25: the last
repetition of a tool that runs its workload four times)."""
import csv, glob, sys, collections
d = sys.argv[1]; frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.25
rows = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[int(len(rows) * (1 - frac)):]
acc = collections.OrderedDict()
for r in rows:
    n = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
    n = n[:n.index("(")] if "(" in n else n
    t = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    a = acc.setdefault(n, [0, 0.0]); a[0] += 1; a[1] += t
span = (int(rows[-1]["End_Timestamp"]) - int(rows[0]["Start_Timestamp"])) / 1e3 if rows else 0
for n, (c, t) in sorted(acc.items(), key=lambda x: -x[1][1]):
    print(f"{t:9.1f} us  x{c:<3d} {n}")
print(f"{sum(t for _, t in acc.values()):9.1f} us  busy of {span:.1f} us span, {len(rows)} dispatches")
if len(sys.argv) > 3 and sys.argv[3] == "seq":      # the dispatches in order: start offset + duration
    t0 = int(rows[0]["Start_Timestamp"])
    for r in rows:
        n = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
        n = n[:n.index("(")] if "(" in n else n
        print(f"{(int(r['Start_Timestamp']) - t0) / 1e3:9.1f} us + {(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3:7.1f} us  {n[:70]}")

"""Kernel sequence of one join in a rocprofv3 kernel_trace.csv (one period between the last two jsum_kernel launches: the tail of one join and the head of the next).
Usage: python tools/jtrace.py <dir-or-csv>"""
import csv, glob, re, sys
p = sys.argv[1]
f = p if p.endswith(".csv") else glob.glob(p + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
ex = [i for i, r in enumerate(rows) if "jsum_kernel" in r["Kernel_Name"]]          # one per join: a full period between two of them
j, k = ex[-2] + 1, ex[-1] + 1
t0 = int(rows[j]["Start_Timestamp"])
agg = {}
for r in rows[j:k]:
    nm = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
    nm = re.sub(r"^void ", "", nm)
    nm = re.sub(r"\(.*", "", nm)[:50]
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%9.1f us +%8.1f us  %s" % ((s - t0) / 1e3, (e - s) / 1e3, nm))
    a = agg.setdefault(nm, [0, 0.0]); a[0] += 1; a[1] += (e - s) / 1e3
print("---- per kernel")
for nm, (c, t) in sorted(agg.items(), key=lambda x: -x[1][1]):
    print("%9.1f us  x%-3d %s" % (t, c, nm))
print("span %.1f us, kernel time %.1f us" % ((int(rows[k - 1]["End_Timestamp"]) - t0) / 1e3, sum(t for c, t in agg.values())))

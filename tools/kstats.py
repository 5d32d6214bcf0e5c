"""Print a rocprofv3 kernel_stats.csv compactly.  Usage: python tools/kstats.py <dir-or-file> [top]"""
import csv, glob, re, sys
p = sys.argv[1]
f = p if p.endswith(".csv") else glob.glob(p + "/**/*kernel_stats.csv", recursive=True)[0]
top = int(sys.argv[2]) if len(sys.argv) > 2 else 12
for r in list(csv.DictReader(open(f)))[:top]:
    m = re.search(r"(\w+_kernel\w*|__amd\w+)", r["Name"])
    name = m.group(1) if m else r["Name"][:30]
    print("%-30s calls=%5s total=%9.3f ms avg=%9.1f us %6s%%" % (name, r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3, r["Percentage"][:6]))

"""HBM bytes and time per kernel of ONE repetition of a tools/op_one.py workload, from its two PMC runs:
    python tools/opmc.py DIR_FETCH_SIZE DIR_WRITE_SIZE [algorithmic_bytes] [reps of op_one.py, default 3]
Each directory is the -d of `rocprofv3 --pmc <COUNTER> --kernel-trace --output-format csv -- python3 tools/op_one.py W`.
The last repetition is found as the longest block of kernel names at the end of the run that repeats the block before it
(op_one.py repeats the statement at least three times; the setup kernels in front do not repeat).  FETCH_SIZE is doubled (gfx950: it reports half of a wide
coalesced read stream, MI355X_MICROARCH.md HBM section), both counters are KiB.  Durations under --pmc are those of a
serialised run (one dispatch at a time): use them as a guide, the trace-only runs for times."""
import collections, csv, glob, sys


def rows_of(d, counter):
    files = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
    if len(files) != 1:
        raise SystemExit(f"{d}: {len(files)} counter_collection.csv files (want exactly one: one directory per run)")
    per = collections.OrderedDict()
    for r in csv.DictReader(open(files[0])):
        if r.get("Counter_Name", counter) != counter:
            continue
        k = int(r["Dispatch_Id"])
        e = per.setdefault(k, {"name": r["Kernel_Name"], "v": 0.0, "t": (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3})
        e["v"] += float(r["Counter_Value"])                   # one row per XCD / dimension instance: summed
    return [per[k] for k in sorted(per)]


def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    return n[:n.index("(")] if "(" in n else n


def last_period(names, reps=3):
    """Dispatches of ONE repetition of a statement that tools/op_one.py ran `reps` times: the longest k such that the last
    reps * k names are reps copies of the last k, which itself is not two equal halves... taken literally that is still
    ambiguous for an even number of repetitions (with reps = 4 the last 2k names "repeat" as well: ADVICE r04), so the answer
    is pinned by the repetition count: the repeating tail is found as the longest block repeated at the end, and ONE repetition
    is that tail's length divided by the number of times the statement ran."""
    reps = max(2, int(reps))
    for k in range(len(names) // reps, 0, -1):
        if all(names[-k:] == names[-(j + 1) * k: len(names) - j * k] for j in range(1, reps)):
            return k
    return len(names)


def main():
    fetch, write = rows_of(sys.argv[1], "FETCH_SIZE"), rows_of(sys.argv[2], "WRITE_SIZE")
    alg = float(sys.argv[3]) if len(sys.argv) > 3 else None
    reps = int(sys.argv[4]) if len(sys.argv) > 4 else 3                       # how often op_one.py ran the statement (its default)
    def one_repetition(rows):
        """Dispatches behind the last marker (op_one.py launches gen_columns_kernel in front of every repetition); traces without a
        marker fall back to the repeating tail."""
        marks = [i for i, r in enumerate(rows) if "gen_columns_kernel" in r["name"]]
        return len(rows) - 1 - marks[-1] if marks else last_period([r["name"] for r in rows], reps)
    kf, kw = one_repetition(fetch), one_repetition(write)
    if kf != kw or [r["name"] for r in fetch[-kf:]] != [r["name"] for r in write[-kw:]]:
        raise SystemExit(f"the two runs do not end in the same kernel sequence (periods {kf} / {kw})")
    acc = collections.OrderedDict()
    for f, w in zip(fetch[-kf:], write[-kw:]):
        a = acc.setdefault(short(f["name"]), [0, 0.0, 0.0, 0.0])
        a[0] += 1; a[1] += 2.0 * f["v"] * 1024.0; a[2] += w["v"] * 1024.0; a[3] += 0.5 * (f["t"] + w["t"])
    tot_r = sum(a[1] for a in acc.values()); tot_w = sum(a[2] for a in acc.values()); tot_t = sum(a[3] for a in acc.values())
    print(f"{'kernel':44s} {'calls':>5s} {'read MB':>10s} {'written MB':>10s} {'us (pmc run)':>12s} {'TB/s':>6s}")
    for n, (c, rd, wr, t) in sorted(acc.items(), key=lambda x: -(x[1][1] + x[1][2])):
        print(f"{n[:44]:44s} {c:5d} {rd / 1e6:10.1f} {wr / 1e6:10.1f} {t:12.1f} {(rd + wr) / max(t, 1e-9) / 1e6:6.2f}")
    line = f"{'TOTAL one repetition, ' + str(kf) + ' dispatches':44s} {'':5s} {tot_r / 1e6:10.1f} {tot_w / 1e6:10.1f} {tot_t:12.1f}"
    if alg:
        line += f"   moved / algorithmic = {(tot_r + tot_w) / alg:.2f}x ({alg / 1e6:.1f} MB algorithmic)"
    print(line)


if __name__ == "__main__":
    main()

"""Copy the judged summaries of ONE tools/evidence.sh run into profiles/<tag>_*.
Usage: python tools/collect_evidence.py r04 gpurun_out/ev_<TAG>
Refuses a source in which any pattern matches more than one file: gpurun merges every call's output into the local
gpurun_out/, so a directory name used for two runs holds two runs' files (a summary stitched from two runs)."""
import collections, csv, glob, json, os, re, shutil, subprocess, sys

tag, src = sys.argv[1], sys.argv[2].rstrip("/")
dst = "profiles"


def only(pattern, required=True):
    f = glob.glob(os.path.join(src, pattern), recursive=True)
    if len(f) > 1:
        raise SystemExit(f"{pattern}: {len(f)} files match in {src} -- a directory holding more than one run; rerun evidence.sh with a fresh TAG")
    if not f and required:
        raise SystemExit(f"{pattern}: nothing matches in {src}")
    return f[0] if f else None


run_id = open(only("run_id.txt")).read()
for name, out in (("bench_plain.json", "bench_latest.json"), ("bench_rocprof.json", "bench_under_rocprof.json"),
                  ("bench_configs_rocprof.json", "bench_configs_under_rocprof.json"), ("ops_bench.log", "ops_bench.log"), ("c5_bench.log", "c5_bench.log")):
    shutil.copy(only(name), os.path.join(dst, f"{tag}_{out}"))
shutil.copy(only("stats/**/*kernel_stats.csv"), os.path.join(dst, f"{tag}_bench_kernel_stats.csv"))
# round 5: bench.py measures the extra configs in a CHILD process, which the profiler follows: the run with configs leaves one
# statistics file per process -- the parent's (headline kernels only) and the child's (every operator pipeline): the one with
# the join's kernels in it is the configs file
cands = glob.glob(os.path.join(src, "stats_configs/**/*kernel_stats.csv"), recursive=True)
with_join = [f for f in cands if "jpart_kernel" in open(f).read()]
if len(with_join) != 1 or len(cands) > 2:
    raise SystemExit(f"stats_configs: {len(cands)} kernel statistics files, {len(with_join)} of them with the join's kernels (want 1 or 2 files, exactly one with)")
shutil.copy(with_join[0], os.path.join(dst, f"{tag}_configs_kernel_stats.csv"))

child = json.loads(open(only("pmc_fetch.json")).read().strip().splitlines()[-1])
assert child.get("pmc_child") is True, "the PMC passes must run bench.py --pmc-child 1 (no setup launch)"
plain = json.loads(open(only("bench_plain.json")).read().strip().splitlines()[-1])
cfg = plain["config"]
out = {"git": subprocess.run(["git", "rev-parse", "--short=12", "HEAD"], capture_output=True, text=True).stdout.strip(), "run_id": run_id,
       "command": "rocprofv3 --pmc <COUNTER> --kernel-trace --output-format csv -- python3 bench.py --pmc-child 1 --steps 2 --warmup 1 --cpu-rows 0 --configs 0 --pmc 0  (one pass per counter; 3 full launches per kernel, no setup launch)",
       "units": "FETCH_SIZE / WRITE_SIZE count KiB; on gfx950 FETCH_SIZE reports half of a wide coalesced read stream, so hbm_read_bytes = 2 * FETCH_SIZE * 1024 (MI355X_MICROARCH.md, HBM section); WRITE_SIZE is exact",
       "kernels": {}, "config": {"rows_per_gpu": cfg["rows_per_gpu"], "groups": cfg["groups"], "chunk_rows": 0,
                                 "producer_launches_per_step": plain["roofline"]["launches"] // plain["steps"]}}
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import headline_launches
for name, ctr in (("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE")):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(only(f"{name}/**/*counter_collection.csv"))):
        m = re.search(r"(fgb_(?:part|agg6|lds)\w*)", r["Kernel_Name"])
        if m and r.get("Counter_Name", ctr) == ctr:
            agg[m.group(1)].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        full = headline_launches(v)
        assert len(full) == len(v) == 3, f"{k}: {len(v)} launches of which {len(full)} full ones (want 3 and 3)"
        out["kernels"].setdefault(k, {})[ctr + "_KiB_per_launch"] = v
        out["kernels"][k][ctr + "_mean_per_launch"] = sum(v) / len(v)
        out["kernels"][k]["launches"] = len(v)
for k, dd in out["kernels"].items():
    dd["hbm_bytes_per_launch_corrected"] = 2 * dd.get("FETCH_SIZE_mean_per_launch", 0) * 1024 + dd.get("WRITE_SIZE_mean_per_launch", 0) * 1024
total = sum(dd["hbm_bytes_per_launch_corrected"] for dd in out["kernels"].values())
out["path_hbm_bytes_per_step"] = total
out["path_over_algorithmic"] = total / (12.0 * cfg["rows_per_gpu"] + 16.0 * cfg["groups"])
out["bench_line_traffic_in_the_same_run"] = plain["roofline"].get("traffic")
json.dump(out, open(os.path.join(dst, f"{tag}_pmc_fgb.json"), "w"), indent=1)
print(json.dumps({k: v["hbm_bytes_per_launch_corrected"] for k, v in out["kernels"].items()}, indent=1), out["path_over_algorithmic"])

# bytes per kernel of the operator workloads (tools/op_one.py under --pmc, one counter per run)
ALG = {"join_c4": 12.0 * (1.25e8 + 1.25e7) + 8.0 * 6.25e7, "join_u32": 12.0 * 1.1e8, "sort20": 16e8, "sort32": 16e8, "sort64": 24e8,
       "sparse_gb": 12e9 + 16.0 * (1 << 20), "sparse_five": 12e9 + 28.0 * (1 << 20), "refgb_hash": 12e8}
with open(os.path.join(dst, f"{tag}_op_traffic.txt"), "w") as fo:
    fo.write(f"# HBM bytes per kernel of one repetition of each operator workload (tools/op_one.py W under rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE,\n"
             f"# one counter per run; tools/opmc.py).  run: {run_id.splitlines()[0]}, git {out['git']}\n")
    for w in ("join_c4", "join_u32", "sort20", "sort32", "sort64", "sparse_gb", "sparse_five", "refgb_hash"):
        df, dw = os.path.join(src, f"opmc_{w}_FETCH_SIZE"), os.path.join(src, f"opmc_{w}_WRITE_SIZE")
        if not (os.path.isdir(df) and os.path.isdir(dw)):
            fo.write(f"\n== {w}: not collected\n")
            continue
        r = subprocess.run([sys.executable, os.path.join(os.path.dirname(__file__), "opmc.py"), df, dw, str(ALG[w])], capture_output=True, text=True)
        logf = os.path.join(src, f"opmc_{w}_FETCH_SIZE.log")                   # the workload's own lines (wall times under the counter run), not the profiler's chatter
        logs = "".join(ln for ln in open(logf) if ln.startswith(w + ":")) if os.path.exists(logf) else ""
        fo.write(f"\n== {w}\n{logs}{r.stdout}{r.stderr}")
print(open(os.path.join(dst, f"{tag}_op_traffic.txt")).read())
# the probes of the same run (round 6: join under skew, the 7-row statements, no-filter producer, SQL stress, SQ counters)
for name, out in (("join_skew.txt", "join_skew.txt"), ("small_latency.txt", "small_latency.txt"), ("nofilter_ab.txt", "nofilter_ab.txt"), ("ingest_bench.log", "ingest_bench.log"),
                  ("strong_rehearsal.txt", "strong_rehearsal.txt"), ("sql_stress.txt", "sql_stress.txt"), ("join_cluster.txt", "join_cluster.txt"), ("groupby_cluster.txt", "groupby_cluster.txt"),
                  ("statement_cluster.txt", "statement_cluster.txt"), ("cluster_traces.txt", "cluster_traces.txt"), ("pmc_join_c4.txt", "pmc_join_c4.txt"), ("pmc_sort64.txt", "pmc_sort64.txt"),
                  ("hashlds.txt", "hashlds.txt"), ("widedigit.txt", "widedigit.txt"), ("hash_pmc_after.txt", "hash_pmc_after.txt"), ("libsort_yardstick.txt", "libsort_yardstick.txt")):
    f = only(name, required=False)
    if f:
        shutil.copy(f, os.path.join(dst, f"{tag}_{out}"))
# ... and the same kernels against the plain-stream floor of their bytes (tools/floor_table.py)
ft = subprocess.run([sys.executable, os.path.join(os.path.dirname(__file__), "floor_table.py"), os.path.join(dst, f"{tag}_op_traffic.txt")], capture_output=True, text=True)
open(os.path.join(dst, f"{tag}_floor_table.md"), "w").write(ft.stdout)

"""Copy the judged summaries of one tools/evidence.sh run from gpurun_out/ev2 into profiles/<tag>_*.
Usage: python tools/collect_evidence.py r02"""
import collections, csv, glob, json, os, re, shutil, sys

tag = sys.argv[1]
src, dst = "gpurun_out/ev3", "profiles"


def first(pattern):
    f = glob.glob(os.path.join(src, pattern), recursive=True)
    return f[0] if f else None


for name, out in (("bench_plain.json", "bench_latest.json"), ("bench_rocprof.json", "bench_under_rocprof.json"),
                  ("ops_bench.log", "ops_bench.log"), ("c5_bench.log", "c5_bench.log")):
    if os.path.exists(os.path.join(src, name)):
        shutil.copy(os.path.join(src, name), os.path.join(dst, f"{tag}_{out}"))
for pat, out in (("stats/**/*kernel_stats.csv", "bench_kernel_stats.csv"), ("stats_configs/**/*kernel_stats.csv", "configs_kernel_stats.csv")):
    f = first(pat)
    if f:
        shutil.copy(f, os.path.join(dst, f"{tag}_{out}"))

bench = json.loads(open(os.path.join(src, "pmc_fetch.json")).read().strip().splitlines()[-1])
cfg = bench["config"]
import subprocess
out = {"git": subprocess.run(["git", "rev-parse", "--short=12", "HEAD"], capture_output=True, text=True).stdout.strip(),
       "command": "rocprofv3 --pmc <COUNTER> --kernel-trace --output-format csv -- python3 bench.py --steps 2 --warmup 1 --cpu-rows 0 --configs 0  (one pass per counter)",
       "units": "FETCH_SIZE / WRITE_SIZE count KiB; on gfx950 FETCH_SIZE reports half of a wide coalesced read stream, so hbm_read_bytes = 2 * FETCH_SIZE * 1024 (MI355X_MICROARCH.md, HBM section); WRITE_SIZE is exact",
       "kernels": {}, "config": {"rows_per_gpu": cfg["rows_per_gpu"], "groups": cfg["groups"], "chunk_rows": 0,
                                 "producer_launches_per_step": bench["roofline"]["launches"] // bench["steps"]}}
for name, ctr in (("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE")):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(first(f"{name}/**/*counter_collection.csv"))):
        m = re.search(r"(fgb_(?:part|agg6|lds)\w*)", r["Kernel_Name"])
        if m:
            agg[m.group(1)].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        out["kernels"].setdefault(k, {})[ctr + "_mean_per_launch"] = sum(v) / len(v)
        out["kernels"][k]["launches"] = len(v)
for k, dd in out["kernels"].items():
    dd["hbm_bytes_per_launch_corrected"] = 2 * dd.get("FETCH_SIZE_mean_per_launch", 0) * 1024 + dd.get("WRITE_SIZE_mean_per_launch", 0) * 1024
json.dump(out, open(os.path.join(dst, f"{tag}_pmc_fgb.json"), "w"), indent=1)
print(json.dumps(out["kernels"], indent=1))

#!/usr/bin/env python3
"""Headline benchmark of BASELINE.json on MI355X.

    python bench.py --gpus N --steps K --warmup W

One step = one pass of the hot path over one batch of synthetic input:
    SELECT k, SUM(v), COUNT(*) FROM t WHERE p > 0.5 GROUP BY k
over three device-resident columns (p f32, k i32, v f32) with 2^20 groups (BASELINE.json configs[2] with the filter;
SURVEY.md 8(d)).  The metric is quoted on ONE 1e9-row table at 1/2/4/8 GPUs ("strong scaling; weak
scaling reported alongside"), so with N > 1 ranks one invocation measures BOTH, one after the other:

  strong  (the line's value / ms_per_step / roofline; "scaling": "strong"; config.rows_total = 1e9): rank r holds rows
          shard_range(1e9, r, N) of the SAME seeded table the N = 1 run holds whole;
  weak    (sub-record "weak": rows_per_gpu = 1e9, rows_total = N x 1e9, its own ms_per_step / value / roofline.frac):
          rank r holds rows [r x 1e9, (r+1) x 1e9).

Either way every rank runs the fused HIP kernels on its shard, the per-GPU partial aggregates (16 B x 2^20) are summed
with an RCCL all-reduce and the merged table is finalised; the start-up measurement (producer geometry x pipelining x
form of the merge) is repeated per mode -- a 0.4 ms strong-scaling step hides no all-reduce that a 3 ms weak step does.
At N = 1 the two coincide: one run, the same line as before.

Run bare with --gpus N > 1 (no torchrun) it starts its own N rank processes.
Setup (before the W warm-up steps): columns generated on the device, plans created, and the path's kernels loaded by
one pass over the first 65536 rows with a small plan of its own (code objects page in on first launch).

Prints ONE JSON line on rank 0 (contract in the task statement) carrying
  roofline      the WHOLE PATH's algorithmic bytes over the step time against the 8 TB/s HBM peak (`frac`), the
                dominant kernel's own figure beside it (HIP events on the launch stream), and what plain stream
                kernels with the byte mix of a two-pass design take in this same process (probes);
  cpu_baseline  the oracle's port of the reference's 32-pass algorithm on one host core over a bounded sample
                (rank 0, N = 1 only);
  configs       (N = 1 only, outside the timed region) the other BASELINE configs and the small-G single-pass
                path, each HIP-event timed (3 warm-ups, median of 10) with its algorithmic bytes and fraction.
                They run in a CHILD process (`--configs-child`, started like the --pmc children: a new process in its
                own process group with a wall-clock budget, never an exec of this one) after this process has its
                headline numbers and has freed its tables; the child writes its JSON after every config, so a child
                that dies, faults or hangs costs the configs it had not reached -- never the headline.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

SEED = 0x4861726B4442
CHILD_PGIDS = []               # process groups of children started here and still running (killed if this process gives up)
HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300-6900 GB/s is what a read stream reaches


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--rows", type=float, default=1e9, help="rows of the table (= rows_total of the strong-scaling run, rows per GPU of the weak one)")
    ap.add_argument("--groups", type=int, default=1 << 20)
    ap.add_argument("--cpu-rows", type=float, default=1.5e8, help="rows of the CPU baseline sample (0 disables)")
    ap.add_argument("--exact", type=int, default=1, help="1: integer-valued v (bit-exact check), 0: uniform [0,1)")
    ap.add_argument("--algo", type=int, default=0)
    ap.add_argument("--chunk-rows", type=int, default=0)
    ap.add_argument("--configs", type=int, default=1, help="0: skip the extra BASELINE-config measurements (N = 1 only)")
    ap.add_argument("--config-scale", type=float, default=1.0, help="scale the rows of the extra configs (tests)")
    ap.add_argument("--configs-budget", type=float, default=300.0, help="wall-clock seconds the configs child may take before it is killed")
    ap.add_argument("--configs-child", type=str, default="", help="internal: measure the extra configs only and write them (incrementally) to this file")
    ap.add_argument("--extras-budget", type=float, default=540.0, help="N = 1: wall-clock seconds for everything after the timed region (probes, checks, PMC children, "
                                                                          "configs child, CPU port); when it runs out the line is printed as it stands")
    ap.add_argument("--first-row", type=int, default=0, help="N = 1: global row number of the table's first row (tests: one rank's shard of a larger table)")
    ap.add_argument("--modes", type=str, default="auto", help="N > 1: which of strong,weak to run (auto = both, strong first)")
    ap.add_argument("--tolerance-check", type=int, default=1, help="1: one untimed pass with the uniform [0,1) value column, held to 1e-5 relative (N = 1)")
    ap.add_argument("--stub", type=int, default=0, help="CPU protocol test: gloo ranks, a numpy step (no GPU, no HIP)")
    ap.add_argument("--pmc", type=int, default=-1, help="HBM traffic of the path's kernels from rocprofv3 --pmc child runs of this script, started "
                                                        "once the headline numbers are in hand (-1: when rocprofv3 is on PATH and N = 1; 0: never; 1: required)")
    ap.add_argument("--pmc-child", type=int, default=0, help="internal: the headline steps only, no probes / checks / configs (run under rocprofv3 --pmc)")
    return ap.parse_args()


def launch_ranks(a):
    """`python bench.py --gpus N` run bare (no torchrun): start N rank processes of this same script, one per GPU,
    with the torchrun environment (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*), and exit with the worst return code.
    The parent has not imported torch or touched HIP, and nothing is exec'd over an initialised process."""
    import socket
    import subprocess
    port = os.environ.get("MASTER_PORT")
    if not port:
        with socket.socket() as so:
            so.bind(("127.0.0.1", 0))
            port = str(so.getsockname()[1])
    procs = []
    for r in range(a.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(a.gpus), LOCAL_WORLD_SIZE=str(a.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=port, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rcs = []
    try:
        for pr in procs:
            rcs.append(pr.wait())
    finally:
        for pr in procs:                        # a rank that died leaves the others waiting in a collective
            if pr.poll() is None:
                try:
                    pr.wait(timeout=30)
                except Exception:
                    pr.kill()
    bad = [rc for rc in rcs if rc]
    raise SystemExit(bad[0] if bad else 0)


def cpu_baseline(rows, G, exact):
    """The reference's algorithm (filter -> materialise -> 32 x 1-bit stable
    split -> head flags -> sequential segmented fold), restated in C, 1 thread."""
    from oracle import oracle as ora
    rows = int(rows)
    p, k, v = ora.gen_columns(SEED, 0, rows, G, bool(exact))
    t0 = time.perf_counter()
    keys, sums, counts = ora.filter_groupby_refalgo_f32(p, k, v, ">", 0.5)
    dt = time.perf_counter() - t0
    # fairness line: a multi-threaded single-pass direct-index aggregate, not the reference's algorithm
    threads = min(os.cpu_count() or 1, 64)
    t1 = time.perf_counter()
    s64, cnt = ora.filter_groupby_dense_f32_mt(p, k, v, ">", 0.5, G, threads)
    dt_mt = time.perf_counter() - t1
    strong = {"value": rows / dt_mt, "unit": "rows/s", "cores": threads, "algorithm": "single-pass direct-index aggregate, OpenMP, private tables (oracle/hark_oracle.c ora_filter_groupby_dense_f32_mt)",
              "agrees_with_port": bool(np.array_equal(cnt[keys.astype(np.int64)], counts.astype(np.int64)))}
    return {"value": rows / dt, "unit": "rows/s", "cores": 1, "kind": "port", "stronger_cpu_baseline": strong,
            "sample": f"first {rows} rows of the same synthetic workload (G={G}), {dt:.2f} s, "
                      f"{int(counts.sum())} survivors, {len(keys)} groups; nproc={os.cpu_count()}",
            "algorithm": "oracle/hark_oracle.c ora_filter_groupby_refalgo_f32 = groupby.fut:8-58 + segmented.fut:7-37"}, (keys, sums, counts)


def under_profiler(env=None):
    """True when this process already runs under rocprofv3 / rocprofiler-sdk (its preload and ROCP_* variables would
    leak into a nested rocprofv3)."""
    env = os.environ if env is None else env
    return any(kk.startswith(("ROCP_", "ROCPROF")) for kk in env) or "rocprof" in env.get("LD_PRELOAD", "")


def headline_launches(values):
    """The counter values of the HEADLINE launches among all launches of one kernel in a profiled child: a launch that
    moved less than half of the largest launch's bytes is not a pass over the N-row workload (e.g. a small setup launch
    that loads the code objects) and must not dilute the per-launch mean."""
    top = max(values)
    return [x for x in values if x >= 0.5 * top]


def measure_traffic(N, G, timeout_s=240, steps=2, warmup=1):
    """HBM bytes per launch of the path's kernels, measured NOW: two child runs of this script's headline steps under
    `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` (one counter per run, with --kernel-trace only, as
    MI355X_MICROARCH.md prescribes), FETCH_SIZE doubled per that guide's gfx950 note.  Children, not an exec: this
    process has initialised the GPU.  The child (--pmc-child) launches the path's kernels for its warm-up and timed
    steps only (no setup launch); every launch counted must be a full pass (headline_launches) and there must be
    exactly steps + warmup of them per kernel and chunk.  Returns (dict, None) or (None, reason)."""
    import collections
    import csv
    import glob
    import re
    import shutil
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3")
    if not exe:
        return None, "rocprofv3 not on PATH"
    if under_profiler():
        return None, "this process runs under a profiler already: no nested rocprofv3"
    tmp = tempfile.mkdtemp(prefix="hark_pmc_", dir="/tmp")
    kernels = collections.defaultdict(dict)
    try:
        for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
            d = os.path.join(tmp, ctr)
            cmd = [exe, "--pmc", ctr, "--kernel-trace", "--output-format", "csv", "-d", d, "--", sys.executable, os.path.abspath(__file__),
                   "--pmc-child", "1", "--rows", str(N), "--groups", str(G), "--steps", str(steps), "--warmup", str(warmup),
                   "--cpu-rows", "0", "--configs", "0", "--pmc", "0"]
            env = {kk: vv for kk, vv in os.environ.items()
                   if not kk.startswith(("ROCP_", "ROCPROF")) and kk not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "HARK_FORCE_PIPELINE", "LD_PRELOAD")}
            env["TMPDIR"] = "/tmp"
            pr = subprocess.Popen(cmd, cwd=tmp, env=env, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, start_new_session=True)
            CHILD_PGIDS.append(pr.pid)
            try:
                _, err = pr.communicate(timeout=timeout_s)
            except subprocess.TimeoutExpired:
                os.killpg(pr.pid, 9)                                  # the exact process group started here
                pr.wait()
                return None, f"rocprofv3 --pmc {ctr} child timed out after {timeout_s} s"
            finally:
                CHILD_PGIDS.remove(pr.pid)
            if pr.returncode:
                return None, f"rocprofv3 --pmc {ctr} child failed (rc {pr.returncode}): {err.decode(errors='replace')[-300:]}"
            files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
            if len(files) != 1:
                return None, f"rocprofv3 --pmc {ctr}: {len(files)} counter_collection.csv files (want exactly one)"
            agg = collections.defaultdict(list)
            for r in csv.DictReader(open(files[0])):
                m = re.search(r"(fgb_(?:part|agg6|agg|lds)\w*)", r["Kernel_Name"])
                if m and r.get("Counter_Name", ctr) == ctr:
                    agg[m.group(1)].append(float(r["Counter_Value"]))
            for kn, vals in agg.items():
                full = headline_launches(vals)
                kernels[kn][ctr + "_KiB_mean_per_launch"] = sum(full) / len(full)
                kernels[kn][ctr + "_launches_profiled"] = len(full)
                kernels[kn][ctr + "_launches_dropped_as_small"] = len(vals) - len(full)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    if not kernels:
        return None, "no fgb_* kernel rows in the counter files"
    for kn, dd in kernels.items():
        lp = {dd.get("FETCH_SIZE_launches_profiled"), dd.get("WRITE_SIZE_launches_profiled")}
        if len(lp) != 1 or None in lp or lp.pop() % (steps + warmup):
            return None, f"{kn}: launches profiled {dd} do not add up to {steps} steps + {warmup} warm-up in both passes"
        dd["launches_profiled"] = dd["FETCH_SIZE_launches_profiled"]
        dd["hbm_read_bytes"] = 2.0 * dd.get("FETCH_SIZE_KiB_mean_per_launch", 0.0) * 1024.0
        dd["hbm_write_bytes"] = dd.get("WRITE_SIZE_KiB_mean_per_launch", 0.0) * 1024.0
        dd["hbm_bytes_per_launch_corrected"] = dd["hbm_read_bytes"] + dd["hbm_write_bytes"]
    return dict(kernels), None


def library_matches_sources():
    """harkdb_amd/_srchash.py: True / False / None (no record) -- is the in-tree libhark.so the build of the sources beside it?"""
    try:
        from harkdb_amd._srchash import library_matches_sources as f
        return f()
    except Exception:
        return None


def build_id():
    """What produced a number: the git commit where a checkout is at hand, else (the GPU box receives a snapshot without
    .git) the first 16 hex digits of sha256 of bench.py and of the loaded libhark.so."""
    import hashlib
    try:
        import subprocess
        h = subprocess.run(["git", "-C", ROOT, "rev-parse", "--short=12", "HEAD"], capture_output=True, text=True, timeout=10).stdout.strip()
        if h:
            return "git " + h
    except Exception:
        pass
    out = []
    for f in (os.path.abspath(__file__), os.environ.get("HARK_LIB") or os.path.join(ROOT, "harkdb_amd", "libhark.so")):
        try:
            out.append(os.path.basename(f) + " sha16 " + hashlib.sha256(open(f, "rb").read()).hexdigest()[:16])
        except Exception:
            pass
    return ", ".join(out) or None


def mode_specs(rows, rank, world, modes="auto", first_row=0):
    """[(label, rows on this rank, first row of this rank, rows of the whole table)] in the order they run.
    N = 1: one run ("single").  N > 1: the metric's own configuration first -- STRONG scaling, the one `rows`-row table
    cut into row ranges (harkdb_amd.dist.shard_range) -- then WEAK scaling (`rows` rows per rank) beside it."""
    from harkdb_amd.dist import shard_range
    rows = int(rows)
    if world == 1:
        return [("single", rows, int(first_row), rows)]
    want = ["strong", "weak"] if modes in ("auto", "", None) else [m.strip() for m in modes.split(",") if m.strip()]
    out = []
    for m in want:
        if m == "strong":
            lo, hi = shard_range(rows, rank, world)
            out.append(("strong", hi - lo, lo, rows))
        elif m == "weak":
            out.append(("weak", rows, rank * rows, rows * world))
        else:
            raise SystemExit(f"--modes: unknown mode {m!r} (strong, weak)")
    return out


def tuning_candidates(env):
    """The start-up measurement's candidates [(producer workgroups, pipelined, form of the merge)], the DEFAULT first.
    HARK_PRODUCER_WGS / HARK_OVERLAP / HARK_ALLREDUCE each pin ONE dimension; the others are still measured (a pinned
    geometry no longer switches the choice of the collective off)."""
    wgs_pin, ov_pin, how_pin = env.get("HARK_PRODUCER_WGS"), env.get("HARK_OVERLAP"), env.get("HARK_ALLREDUCE")
    if wgs_pin is not None:
        geo = [(int(wgs_pin), True), (int(wgs_pin), False)]
    else:
        geo = [(240, True), (0, True), (0, False)]
    if ov_pin is not None:
        geo = [g for g in geo if g[1] == (ov_pin != "0")] or [(int(wgs_pin or 0), ov_pin != "0")]
    hows = [how_pin] if how_pin else ["allreduce", "rs_ag"]
    return [(w, o, h) for h in hows for (w, o) in geo]


def candidate_name(wgs, overlap, how):
    return ("%d workgroups" % wgs if wgs else "all CUs") + (", pipelined" if overlap else ", serial") + ", " + how


TUNE_MARGIN = 0.03


def pick_candidate(times):
    """times: {name: ms} in candidate order (the default first).  The default stays unless another candidate beats it by
    more than TUNE_MARGIN: six perf_counter-timed steps with barriers cannot tell near-ties apart, and a headline whose
    geometry flips between runs is not comparable run to run.  Returns (name, margin of the choice over the default)."""
    names = list(times)
    default = names[0]
    best = min(names, key=lambda nm: times[nm])
    if best != default and times[best] < (1.0 - TUNE_MARGIN) * times[default]:
        return best, 1.0 - times[best] / times[default]
    return default, 0.0


def stub_columns(first_row, n, G):
    """Counter-based columns of the stub (a function of the GLOBAL row index only: every world size sees the same table)."""
    i = np.arange(first_row, first_row + n, dtype=np.uint64)
    h = (i + np.uint64(0x9E3779B97F4A7C15)) * np.uint64(0xBF58476D1CE4E5B9)
    h ^= h >> np.uint64(31)
    h *= np.uint64(0x94D049BB133111EB)
    h ^= h >> np.uint64(29)
    return ((h >> np.uint64(20)) & np.uint64(0xFFFFFF)).astype(np.float64) / 2.0**24, (h % np.uint64(G)).astype(np.int64)


def stub_main(a):
    """The rank protocol of main() on CPU (tests/test_bench_launcher.py): gloo ranks, a numpy aggregate as the step,
    the same modes (strong first, weak beside it), barrier / MAX-over-ranks timing and JSON shape.  Measures nothing."""
    import torch
    import torch.distributed as dist
    from harkdb_amd import dist as hd
    os.environ.setdefault("HARK_DIST_BACKEND", "gloo")
    rank, local, world = hd.init_process_group("cpu")
    if world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}")
    G = int(a.groups)
    records = []
    for label, n_local, first_row, rows_total in mode_specs(a.rows, rank, world, a.modes):
        p, k = stub_columns(first_row, n_local, G)
        sums = torch.zeros(G, dtype=torch.float64)
        cnts = torch.zeros(G, dtype=torch.int64)

        def step():
            keep = p > 0.5
            sums.copy_(torch.from_numpy(np.bincount(k[keep], weights=np.ones(int(keep.sum())), minlength=G)))
            cnts.copy_(torch.from_numpy(np.bincount(k[keep], minlength=G)))
            hd.allreduce_partials(sums, cnts, how="allreduce")

        for _ in range(a.warmup):
            step()
        if dist.is_initialized():
            dist.barrier()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            step()
        if dist.is_initialized():
            dist.barrier()
        t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
        per_rank = [torch.zeros_like(t) for _ in range(world)]
        if dist.is_initialized():
            dist.all_gather(per_rank, t)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        else:
            per_rank = [t.clone()]
        total = torch.tensor([int((p > 0.5).sum()), n_local], dtype=torch.int64)
        if dist.is_initialized():
            dist.all_reduce(total)
        el = float(t.item())
        records.append({"mode": label, "rows_total": rows_total, "rows_local": n_local, "elapsed": el, "survivors": int(total[0].item()),
                        "rows_seen_by_all_ranks": int(total[1].item()), "per_rank": [float(x.item()) for x in per_rank],
                        "count_checksum": int(cnts.sum().item()) == int(total[0].item())})
    if rank == 0:
        def rec(r):
            return {"value": r["rows_total"] / (r["elapsed"] / a.steps), "unit": "rows/s", "ms_per_step": r["elapsed"] / a.steps * 1e3,
                    "scaling": "weak" if r["mode"] == "weak" else "strong", "ms_per_step_by_rank": [x / a.steps * 1e3 for x in r["per_rank"]],
                    "config": {"workload": "stub (CPU protocol test)", "rows_total": r["rows_total"], "rows_per_gpu": r["rows_total"] // world,
                               "rows_this_rank": r["rows_local"]},
                    "check": {"count_checksum": r["count_checksum"], "rows_seen_by_all_ranks": r["rows_seen_by_all_ranks"], "survivors": r["survivors"]}}
        head = rec(records[0])
        line = {"metric": "stub", "value": head["value"], "unit": "rows/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
                "ms_per_step": head["ms_per_step"], "higher_is_better": True, "scaling": head["scaling"], "vs_baseline": None, "dtype": "f32",
                "data": "synthetic", "config": head["config"], "rccl_ranks": dist.get_world_size() if dist.is_initialized() else 1,
                "library_linked_from_the_sources_in_the_tree": library_matches_sources() if not os.environ.get("HARK_LIB") else None,
                "backend": dist.get_backend() if dist.is_initialized() else None, "ms_per_step_by_rank": head["ms_per_step_by_rank"],
                "check": head["check"], "weak": None}
        for r in records[1:]:
            line[r["mode"]] = rec(r)
        print(json.dumps(line), flush=True)
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


def event_ms(torch, fn, warm=3, reps=10):
    """Median HIP-event time of fn() on torch's current stream (= the engine's stream after share_stream)."""
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    ts.sort()
    return ts[len(ts) // 2]


# ---- the operator workloads behind `configs.*`, one builder each: bench.py times them with HIP events, tools/op_one.py
#      runs the SAME builders under rocprofv3 (kernel traces, PMC byte counters).  A builder returns
#      {"run": () -> Result, "bytes": (Result) -> algorithmic bytes, "rows": rows read, "info": {...}, "keep": device data}.
SPARSE_MUL = 2654435761                                       # odd: g -> g * SPARSE_MUL mod 2^32 is a bijection of the u32 keys


def _gen_i32(torch, eng, dev, seed, n, G):
    t = torch.empty((n + 3) // 4 * 4, dtype=torch.int32, device=dev)
    eng.gen_columns(seed, 0, n, G, True, None, t.data_ptr(), None)
    return t[:n]


def w_c2(torch, eng, dev, scale=1.0, want="c2"):
    n2 = int(1e8 * scale) // 4 * 4
    cols = [torch.empty(n2, dtype=torch.float32, device=dev) for _ in range(8)]
    for j in range(0, 8, 2):
        eng.gen_columns(SEED + j, 0, n2, 1 << 20, False, cols[j].data_ptr(), None, cols[j + 1].data_ptr())
    t8 = eng.table_from_device(n2, [c.data_ptr() for c in cols], [np.float32] * 8, keepalive=cols)
    if want == "c1":
        return {"run": lambda: eng.query_sel(t8, [0, 2]), "bytes": lambda r: 16.0 * n2, "rows": n2, "keep": (t8, cols),
                "info": {"statement": "SELECT c0,c2 FROM t8 (query_sel, main.fut:7)"}}
    return {"run": lambda: eng.filter_sel(t8, 1, ">", 0.5, [0, 2], want_row_index=True), "bytes": lambda r: 12.0 * n2 + 16.0 * r.shape[0], "rows": n2,
            "keep": (t8, cols), "info": {"statement": "SELECT rowid,c0,c2 FROM t8 WHERE c1>0.5 (8 f32 columns resident)",
                                         "bytes_model": "read c0,c1,c2 (12 B/row) + write 2 x f32 + i64 row index per survivor"}}


def w_c1(torch, eng, dev, scale=1.0):
    return w_c2(torch, eng, dev, scale, want="c1")


def w_refgb(torch, eng, dev, scale=1.0):
    n8 = int(1e8 * scale) // 4 * 4
    ku, au = _gen_i32(torch, eng, dev, SEED, n8, 1 << 20), _gen_i32(torch, eng, dev, SEED + 9, n8, 1 << 16)
    tu = eng.table_from_device(n8, [ku.data_ptr(), au.data_ptr()], [np.uint32, np.uint32], keepalive=(ku, au))
    return {"run": lambda: eng.query_groupby(tu, 0, [1, 1], [2, 3]), "bytes": lambda r: 12.0 * n8, "rows": n8, "keep": (tu,),
            "info": {"statement": "query_groupby(db, 0, [1, 1], [sum, max]) (main.fut:9), 2^20 dense keys",
                     "note": "one statistics pass (sum + max of ONE column: the column is carried once)"}}


def w_refgb_hash(torch, eng, dev, scale=1.0):
    n8 = int(1e8 * scale) // 4 * 4
    kd, au = _gen_i32(torch, eng, dev, SEED + 13, n8, 1 << 21), _gen_i32(torch, eng, dev, SEED + 9, n8, 1 << 16)
    ks = (kd.to(torch.int64) * SPARSE_MUL).to(torch.int32)     # 2^21 distinct keys spread over the whole u32 range
    del kd
    th = eng.table_from_device(n8, [ks.data_ptr(), au.data_ptr()], [np.uint32, np.uint32], keepalive=(ks, au))
    return {"run": lambda: eng.query_groupby(th, 0, [1, 1], [2, 3]), "bytes": lambda r: 12.0 * n8, "rows": n8, "keep": (th,),
            "info": {"statement": "query_groupby(db, 0, [1, 1], [sum, max]) (main.fut:9), 2^21 distinct u32 keys spread over [0, 2^32)"}}


def w_refgb_hash1(torch, eng, dev, scale=1.0):
    w = w_refgb_hash(torch, eng, dev, scale)
    th, n8 = w["keep"][0], w["rows"]
    return {"run": lambda: eng.query_groupby(th, 0, [1], [2]), "bytes": lambda r: 8.0 * n8, "rows": n8, "keep": (th,),
            "info": {"statement": "query_groupby(db, 0, [1], [sum]) (main.fut:9), 2^21 distinct u32 keys spread over [0, 2^32)"}}


def w_sort(torch, eng, dev, scale=1.0, bits=20):
    n8 = int(1e8 * scale) // 4 * 4
    au = _gen_i32(torch, eng, dev, SEED + 9, n8, 1 << 16)
    if bits == 64:
        g = torch.Generator(device=dev)
        g.manual_seed(1)
        keys = torch.randint(-2**62, 2**62, (n8,), dtype=torch.int64, device=dev, generator=g)
        t = eng.table_from_device(n8, [keys.data_ptr(), au.data_ptr()], [np.int64, np.int32], keepalive=(keys, au))
        return {"run": lambda: eng.sort(t, 0, [0, 1]), "bytes": lambda r: 24.0 * n8, "rows": n8, "keep": (t,),
                "info": {"statement": "ORDER BY an i64 key spread over 63 bits, key + one column out (16-byte tuple sort)"}}
    ku = _gen_i32(torch, eng, dev, SEED, n8, (1 << bits) if bits < 31 else 0x7FFFFFFF)
    t = eng.table_from_device(n8, [ku.data_ptr(), au.data_ptr()], [np.uint32, np.uint32], keepalive=(ku, au))
    return {"run": lambda: eng.sort(t, 0, [0, 1]), "bytes": lambda r: 16.0 * n8, "rows": n8, "keep": (t,),
            "info": {"statement": f"ORDER BY a {bits}-bit u32 key, key + one column out (stable LSD radix sort)"}}


def w_join_u32(torch, eng, dev, scale=1.0):
    n8 = int(1e8 * scale) // 4 * 4
    m8 = n8 // 10
    au = _gen_i32(torch, eng, dev, SEED + 9, n8, 1 << 16)
    kb8, vb8, kp8 = _gen_i32(torch, eng, dev, SEED + 3, m8, 1 << 30), _gen_i32(torch, eng, dev, SEED + 4, m8, 1 << 16), _gen_i32(torch, eng, dev, SEED + 5, n8, 1 << 30)
    tp8 = eng.table_from_device(n8, [kp8.data_ptr(), au.data_ptr()], [np.uint32, np.uint32], keepalive=(kp8, au))
    tb8 = eng.table_from_device(m8, [kb8.data_ptr(), vb8.data_ptr()], [np.uint32, np.uint32], keepalive=(kb8, vb8))
    sb = torch.sort(kb8).values                                                      # pairs a join must find: every probe key's partners in the build side
    hits = 0
    for c0 in range(0, n8, 1 << 25):
        ch = kp8[c0: c0 + (1 << 25)]
        hits += int((torch.searchsorted(sb, ch, right=True) - torch.searchsorted(sb, ch, right=False)).sum().item())
    del sb
    return {"run": lambda: eng.join(tp8, tb8, 0, 0, [0, 1], [1]), "bytes": lambda r: 12.0 * (n8 + m8) + 12.0 * r.shape[0], "rows": n8 + m8, "keep": (tp8, tb8),
            "info": {"probe_rows": n8, "build_rows": m8, "pairs_expected": hits,
                     "statement": "join(db1, db2, 0, 0, [0, 1], [1]) (join.fut:52) on u32 keys, ~10 % of the probe rows match"}}


def w_join_u32_sorted(torch, eng, dev, scale=1.0):
    """The reference's join with the PROBE table kept in key order (a fact table sorted by its foreign key): the same columns as
    w_join_u32, the probe rows sorted by the key -- the search path of k_cjoin.hip instead of the partition."""
    w = w_join_u32(torch, eng, dev, scale)
    tp8, tb8 = w["keep"]
    n8 = w["info"]["probe_rows"]
    from harkdb_amd.dist import tensor_from_ptr
    kp = tensor_from_ptr(tp8.device_ptr(0), n8, np.int32, dev)
    srt = kp.sort().values                                                           # all keys below 2^30: signed order = the join's unsigned order
    a2 = tensor_from_ptr(tp8.device_ptr(1), n8, np.int32, dev).clone()
    torch.cuda.synchronize()
    ts = eng.table_from_device(n8, [srt.data_ptr(), a2.data_ptr()], [np.uint32, np.uint32], keepalive=(srt, a2))
    tp8.free()
    info = dict(w["info"], statement="join(db1, db2, 0, 0, [0, 1], [1]) (join.fut:52) on u32 keys, db1 SORTED by its key column, ~10 % of its rows match")
    return {"run": lambda: eng.join(ts, tb8, 0, 0, [0, 1], [1]), "bytes": w["bytes"], "rows": w["rows"], "keep": (ts, tb8), "info": info}


def w_sort20_sorted(torch, eng, dev, scale=1.0):
    """ORDER BY a u32 key column that is in order already: the sort's first read notices and no radix pass runs."""
    w = w_sort(torch, eng, dev, scale, bits=20)
    t, n8 = w["keep"][0], w["rows"]
    from harkdb_amd.dist import tensor_from_ptr
    ks = tensor_from_ptr(t.device_ptr(0), n8, np.int32, dev).sort().values
    a2 = tensor_from_ptr(t.device_ptr(1), n8, np.int32, dev).clone()
    torch.cuda.synchronize()
    t2 = eng.table_from_device(n8, [ks.data_ptr(), a2.data_ptr()], [np.uint32, np.uint32], keepalive=(ks, a2))
    t.free()
    return {"run": lambda: eng.sort(t2, 0, [0, 1]), "bytes": w["bytes"], "rows": n8, "keep": (t2,),
            "info": {"statement": "ORDER BY a 20-bit u32 key that is in order already, key + one column out (no radix pass runs)"}}


def w_join_c4(torch, eng, dev, scale=1.0):
    n4, s4 = int(1.25e8 * scale), int(1.25e7 * scale)
    mul = -7046029254386353131                                                       # 0x9E3779B97F4A7C15 as i64: odd, so i -> i*mul is a bijection mod 2^64
    bk = torch.arange(s4, dtype=torch.int64, device=dev) * mul                       # unique build keys spread over 64 bits
    j = _gen_i32(torch, eng, dev, SEED + 21, n4, 2 * s4)                             # half of the probe rows find a partner
    pk = j.to(torch.int64) * mul
    prow, brow = torch.arange(n4, dtype=torch.int32, device=dev), torch.arange(s4, dtype=torch.int32, device=dev)
    hits = int((j < s4).sum().item())
    del j
    tp = eng.table_from_device(n4, [pk.data_ptr(), prow.data_ptr()], [np.int64, np.int32], keepalive=(pk, prow))
    tb = eng.table_from_device(s4, [bk.data_ptr(), brow.data_ptr()], [np.int64, np.int32], keepalive=(bk, brow))
    return {"run": lambda: eng.join(tp, tb, 0, 0, [1], [1]), "bytes": lambda r: 12.0 * (n4 + s4) + 8.0 * hits, "rows": n4 + s4, "keep": (tp, tb),
            "info": {"probe_rows": n4, "build_rows": s4, "pairs_expected": hits,
                     "statement": "probe JOIN build ON i64 key -> (probe row id, build row id), reference order (key, left row, right row)",
                     "note": "1/8 of configs[3]; over 8 GPUs the rows arrive by an all-to-all first (xGMI-bound, DESIGN.md 6)"}}


def w_sparse_gb(torch, eng, dev, scale=1.0, cols=None, G=1 << 20):
    """The headline statement over SPARSE i32 keys: the 2^20 dense keys g of the headline table mapped to g * SPARSE_MUL mod
    2^32 (a bijection), i.e. 2^20 distinct keys spread over [-2^31, 2^31) -- no key is a slot index any more."""
    N = int(1e9 * scale) // 4 * 4
    if cols is None:
        p, k, v = (torch.empty(N, dtype=dt, device=dev) for dt in (torch.float32, torch.int32, torch.float32))
        eng.gen_columns(SEED, 0, N, G, True, p.data_ptr(), k.data_ptr(), v.data_ptr())
    else:
        p, k, v = cols
        N = p.numel()
    ks = torch.empty(N, dtype=torch.int32, device=dev)
    step = 1 << 27
    for lo in range(0, N, step):                               # piecewise: the i64 intermediate stays small
        ks[lo:lo + step] = (k[lo:lo + step].to(torch.int64) * SPARSE_MUL).to(torch.int32)
    ts = eng.table_from_device(N, [p.data_ptr(), ks.data_ptr(), v.data_ptr()], [np.float32, np.int32, np.float32], keepalive=(p, ks, v))
    return {"run": lambda: eng.filter_groupby(ts, [(0, ">", 0.5)], 1, [("sum", 2), ("count", 0)]), "bytes": lambda r: 12.0 * N + 16.0 * r.shape[0], "rows": N,
            "keep": (ts, p, k, v, ks),
            "info": {"statement": "SELECT k,SUM(v),COUNT(*) FROM t WHERE p>0.5 GROUP BY k -- k: 2^20 distinct i32 keys spread over [-2^31, 2^31)",
                     "groups": G}}


def w_sparse_five(torch, eng, dev, scale=1.0):
    """SUM, MAX, MIN, AVG of the value column + COUNT over the sparse keys of w_sparse_gb: one hash producer, ONE statistics consumer pass."""
    w = w_sparse_gb(torch, eng, dev, scale)
    ts, N = w["keep"][0], w["rows"]
    five = [("sum", 2), ("max", 2), ("min", 2), ("avg", 2), ("count", 0)]
    return {"run": lambda: eng.filter_groupby(ts, [(0, ">", 0.5)], 1, five), "bytes": lambda r: 12.0 * N + 28.0 * r.shape[0], "rows": N, "keep": w["keep"],
            "info": {"statement": "SELECT k,SUM(v),MAX(v),MIN(v),AVG(v),COUNT(*) FROM t WHERE p>0.5 GROUP BY k -- 2^20 sparse i32 keys", "groups": w["info"]["groups"]}}


WORKLOADS = {"sparse_five": w_sparse_five, "c1": w_c1, "c2": w_c2, "refgb": w_refgb, "refgb_hash": w_refgb_hash, "refgb_hash1": w_refgb_hash1, "join_u32": w_join_u32, "join_u32_sorted": w_join_u32_sorted, "sort20_sorted": w_sort20_sorted, "join_c4": w_join_c4, "sparse_gb": w_sparse_gb,
             "sort20": lambda *a: w_sort(*a, bits=20), "sort32": lambda *a: w_sort(*a, bits=31), "sort64": lambda *a: w_sort(*a, bits=64)}


def reference_csv_latency(reps=300):
    """configs[0]: `select col1, col3` / `select col1, max(col3) ... group by col1` on tests/golden/data.csv through the Python
    surface, one launch + one synchronisation each (harkdb_amd/csrc/k_small.hip); the oracle's entries timed on the same rows."""
    from harkdb_amd import FutharkContext
    from oracle import oracle as ora
    fc = FutharkContext()
    data = os.path.join(ROOT, "tests", "golden", "data.csv")
    fc.create_table("game_1", data)
    db = np.loadtxt(data, delimiter=",", skiprows=1, dtype=np.int64)
    res = {}
    for name, stmt, cpu in (("query_sel", "select col1, col3 from game_1", lambda: ora.query_sel(db, [0, 2])),
                            ("query_groupby", "select col1,  max(col3) from game_1 group by col1", lambda: ora.query_groupby(db, 0, [0, 2], [0, 3]))):
        got = fc.sql(stmt)
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter(); fc.sql(stmt); ts.append(time.perf_counter() - t0)
        ts.sort()
        want = cpu()
        tc = []
        for _ in range(reps):
            t0 = time.perf_counter(); cpu(); tc.append(time.perf_counter() - t0)
        tc.sort()
        res[name] = {"ms": ts[len(ts) // 2] * 1e3, "us_median": ts[len(ts) // 2] * 1e6, "us_min": ts[0] * 1e6, "statement": stmt, "rows": int(db.shape[0]),
                     "equals_cpu_port": bool(np.array_equal(np.asarray(got).astype(np.int64), np.asarray(want).astype(np.int64))),
                     "path": fc.FutEnv.last_groupby_path() if name == "query_groupby" else "small (one launch)",
                     "cpu_baseline": {"value": tc[len(tc) // 2] * 1e6, "unit": "us per statement (median)", "cores": 1, "kind": "port",
                                      "sample": "the same 7 rows: oracle/hark_oracle.c ora_" + name + " through its ctypes wrapper"}}
    res["note"] = ("wall clock through FutharkContext.sql(): cached plan, ONE kernel launch, ONE stream synchronisation, the result matrix "
                   "written by the kernel into a pinned host block; launch-latency bound, no roofline applies")
    return res


def extra_configs(torch, eng, dev, a, sink, out):
    """The other BASELINE configs on one GPU (or one GPU's share of them) and the small-G single-pass path.
    Outside the timed region, in the configs child; every number HIP-event timed on the launch stream, 3 warm-ups, median
    of 10.  `out` is filled config by config and sink() called after each (the child's incremental result file)."""
    from harkdb_amd.engine import FgbPlan
    N = int(a.rows)
    p = torch.empty(N, dtype=torch.float32, device=dev)
    k = torch.empty(N, dtype=torch.int32, device=dev)
    v = torch.empty(N, dtype=torch.float32, device=dev)
    eng.gen_columns(SEED, 0, N, int(a.groups), bool(a.exact), p.data_ptr(), k.data_ptr(), v.data_ptr())

    def entry(ms, alg_bytes, rows, **kw):
        d = {"ms": ms, "algorithmic_bytes": alg_bytes, "achieved_GBps": alg_bytes / (ms * 1e-3) / 1e9,
             "frac_of_peak": alg_bytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "rows": rows, "rows_per_s": rows / (ms * 1e-3)}
        d.update(kw)
        return d

    def timed(w, warm=2, reps=5, **kw):
        """HIP-event median of a WORKLOADS builder's statement; its result shape and algorithmic bytes from the last run."""
        last = [None, None]

        def fn():
            r = w["run"]()
            last[0], last[1] = r.shape, w["bytes"](r)
            r.free()

        ms = event_ms(torch, fn, warm=warm, reps=reps)
        return entry(ms, last[1], w["rows"], result_shape=list(last[0]), **w["info"], **kw)

    # ---- BASELINE configs[2] AS WRITTEN: GROUP BY i32 key with SUM/COUNT, no filter (groupby.fut:51-62 has none either):
    #      8 B/row algorithmic (SURVEY.md 8(d) "C3 without filter"), every row a survivor -- the far end of the selectivity envelope
    G = int(a.groups)
    so, co = torch.empty(G, dtype=torch.float32, device=dev), torch.empty(G, dtype=torch.int64, device=dev)
    plan = FgbPlan(eng, N, G, timing=1)

    def step_nf():
        plan.reset()
        plan.run(None, ">", 0.0, k.data_ptr(), v.data_ptr(), N)
        plan.finish(so.data_ptr(), co.data_ptr())

    ms = event_ms(torch, step_nf)
    kms, kl = plan.timing()
    out["C3_no_filter"] = entry(ms, 8.0 * N + 16.0 * G, N, groups=G, statement="SELECT k,SUM(v),COUNT(*) FROM t GROUP BY k  (configs[2] as written, p = NULL at the C ABI)",
                                kernel_ms={kk: kms[kk] / max(1, kl[kk]) for kk in kms if kl[kk]}, count_checksum=int(co.sum().item()) == N,
                                sum_checksum=(float(so.to(torch.float64).sum().item()) == float(v.to(torch.float64).sum().item())) if a.exact else None,
                                bytes_model="read k, v (8 B/row) + 16 B per group; the partition adds 6 B written + 6 B read per row (20 B/row moved)")
    plan.free()
    sink()

    # ---- the same two statements on the table SORTED by k (a table kept in key order): the partition's rings are sized for keys that
    #      scatter -- consecutive rows of one key all fall into one ring -- so such columns take the window path (fgb_window_kernel:
    #      every workgroup aggregates a contiguous stretch of rows in an LDS window of consecutive keys; one pass, no pairs written).
    #      Only k is sorted (p and v keep their rows): the survivors and the grand totals stay those of the headline.
    try:
        ksorted = k.sort().values
        torch.cuda.synchronize()
        for name, pp, nbytes, stmt in (("HEADLINE_sorted_keys", p.data_ptr(), 12.0 * N + 16.0 * G, "SELECT k,SUM(v),COUNT(*) FROM t WHERE p>0.5 GROUP BY k -- t sorted by k"),
                                       ("C3_no_filter_sorted_keys", None, 8.0 * N + 16.0 * G, "SELECT k,SUM(v),COUNT(*) FROM t GROUP BY k -- t sorted by k")):
            plan = FgbPlan(eng, N, G, timing=1)

            def step_sorted():
                plan.reset()
                plan.run(pp, ">", 0.5, ksorted.data_ptr(), v.data_ptr(), N)
                plan.finish(so.data_ptr(), co.data_ptr())

            ms = event_ms(torch, step_sorted)
            kms, kl = plan.timing()
            want = int((p > 0.5).sum().item()) if pp is not None else N
            out[name] = entry(ms, nbytes, N, groups=G, statement=stmt, path="window (one pass)" if kl.get("consumer", 0) == 0 else "partition + LDS consumer",
                              kernel_ms={kk: kms[kk] / max(1, kl[kk]) for kk in kms if kl[kk]}, count_checksum=int(co.sum().item()) == want)
            plan.free()
            sink()
        del ksorted
    except Exception as e:
        out["HEADLINE_sorted_keys"] = {"error": repr(e)}
    del so, co
    torch.cuda.empty_cache()
    sink()

    # ---- BASELINE configs[0]: the reference's own two statements on its 7-row data.csv (README.md:42, test.py:7), end to end
    #      through FutharkContext.sql() -- wall clock, median of 300 -- with the CPU restatement of the same entries on the same
    #      seven rows beside it (the sequential-C path this build replaces needs microseconds there: the launch + one
    #      synchronisation of the small-table path are what is left to compare)
    try:
        out["C0_reference_csv"] = reference_csv_latency()
    except Exception as e:
        out["C0_reference_csv"] = {"error": repr(e)}
    sink()

    # ---- the headline statement at small G (single-pass LDS path while 12 B x G fits a workgroup's LDS)
    for name, G in (("G16", 16), ("G4096", 4096), ("G13000", 13000)):
        eng.gen_columns(SEED, 0, N, G, bool(a.exact), None, k.data_ptr(), None)
        so, co = torch.empty(G, dtype=torch.float32, device=dev), torch.empty(G, dtype=torch.int64, device=dev)
        plan = FgbPlan(eng, N, G, timing=1)

        def step():
            plan.reset()
            plan.run(p.data_ptr(), ">", 0.5, k.data_ptr(), v.data_ptr(), N)
            plan.finish(so.data_ptr(), co.data_ptr())

        ms = event_ms(torch, step)
        kms, kl = plan.timing()
        dom = max(kms, key=lambda kk: kms[kk])
        ok = int(co.sum().item()) == int((p > 0.5).sum().item())
        out[name] = entry(ms, 12.0 * N + 16.0 * G, N, groups=G, path={"single": "fgb_lds_kernel (one pass)", "producer": "partition + LDS consumer"}.get(dom, dom),
                          kernel_ms=sum(kms.values()) / max(1, kl[dom]), count_checksum=ok, statement="SELECT k,SUM(v),COUNT(*) WHERE p>0.5 GROUP BY k")
        plan.free()
        sink()
    del so, co

    # ---- the envelope of the headline statement: selectivity x G (partition bytes scale with the selectivity; the
    #      single-pass path ends at G = 13 568).  p is uniform in [0, 1): `p > thr` keeps 1 - thr of the rows.
    sweep = {}
    for G, thr in ((1 << 14, 0.5), (1 << 16, 0.5), (1 << 18, 0.5), (1 << 20, 0.9), (1 << 20, 0.5), (1 << 20, 0.1), (1 << 20, 0.0)):
        eng.gen_columns(SEED, 0, N, G, bool(a.exact), None, k.data_ptr(), None)
        so, co = torch.empty(G, dtype=torch.float32, device=dev), torch.empty(G, dtype=torch.int64, device=dev)
        plan = FgbPlan(eng, N, G, timing=1)

        def step():
            plan.reset()
            plan.run(p.data_ptr(), ">=" if thr == 0.0 else ">", thr, k.data_ptr(), v.data_ptr(), N)   # p >= 0: every row survives, the predicate column is still read
            plan.finish(so.data_ptr(), co.data_ptr())

        ms = event_ms(torch, step)
        kms, kl = plan.timing()
        surv = int(co.sum().item())
        sweep[f"G2^{G.bit_length() - 1}_sel{1.0 - thr:.1f}"] = entry(
            ms, 12.0 * N + 16.0 * G, N, groups=G, selectivity=surv / N, survivors=surv,
            kernel_ms={kk: kms[kk] / max(1, kl[kk]) for kk in kms if kl[kk]}, count_checksum=surv == int(((p >= thr) if thr == 0.0 else (p > thr)).sum().item()))
        plan.free()
        del so, co
        out["SWEEP_selectivity_x_groups"] = sweep
        sink()

    # ---- the headline statement over SPARSE keys (arbitrary i32 keys: hash partition + LDS hash tables, not slot indices),
    #      through the SQL entry hark_entry_filter_groupby; checked against the dense result of the same entry after mapping
    #      the keys (bit-exact: integer-valued f32 sums, counts)
    G = int(a.groups)
    eng.gen_columns(SEED, 0, N, G, bool(a.exact), None, k.data_ptr(), None)          # the headline key column again (the sweep left G = 2^20 / other G)
    td = eng.table_from_device(N, [p.data_ptr(), k.data_ptr(), v.data_ptr()], [np.float32, np.int32, np.float32], keepalive=(p, k, v))
    w = w_sparse_gb(torch, eng, dev, cols=(p, k, v), G=G)
    rs = w["run"]()
    path = eng.last_groupby_path()
    rd = eng.filter_groupby(td, [(0, ">", 0.5)], 1, [("sum", 2), ("count", 0)])
    dk, dsum, dcnt = (rd.column(j) for j in range(3))
    sk, ssum, scnt = (rs.column(j) for j in range(3))
    mapped = (dk.astype(np.int64) * SPARSE_MUL).astype(np.uint32).view(np.int32)
    o = np.argsort(mapped, kind="stable")
    same = bool(np.array_equal(sk, mapped[o]) and np.array_equal(scnt, dcnt[o]) and
                (np.array_equal(ssum, dsum[o]) if a.exact else np.allclose(ssum, dsum[o], rtol=1e-5, atol=0)))
    rs.free()
    rd.free()
    dense_ms = event_ms(torch, lambda: eng.filter_groupby(td, [(0, ">", 0.5)], 1, [("sum", 2), ("count", 0)]).free(), warm=1, reps=5)
    out["SPARSE_groupby"] = timed(w, warm=1, reps=5, path=path, equals_dense_result_after_key_mapping=same,
                                  dense_keys_same_entry_ms=dense_ms, note="same rows, same entry (hark_entry_filter_groupby_and, result allocation and "
                                  "group-set read-out included in both); keys = dense key x 2654435761 mod 2^32 as i32")
    out["SPARSE_groupby"]["over_dense_same_entry"] = out["SPARSE_groupby"]["ms"] / dense_ms
    sink()
    # five aggregates of ONE column over the same sparse keys: one hash producer, one statistics consumer pass, one sort
    ts = w["keep"][0]
    five = [("sum", 2), ("max", 2), ("min", 2), ("avg", 2), ("count", 0)]
    out["SPARSE_five_aggregates"] = timed({"run": lambda: eng.filter_groupby(ts, [(0, ">", 0.5)], 1, five), "bytes": lambda r: 12.0 * N + 28.0 * r.shape[0], "rows": N,
                                           "info": {"statement": "SELECT k,SUM(v),MAX(v),MIN(v),AVG(v),COUNT(*) FROM t WHERE p>0.5 GROUP BY k -- the sparse keys of SPARSE_groupby",
                                                    "groups": G}}, warm=1, reps=5)
    out["SPARSE_five_aggregates"]["path"] = eng.last_groupby_path()
    sink()
    td.free()
    del w, td, ts, p, k, v                                    # the 1e9-row columns are not needed below
    torch.cuda.empty_cache()

    # ---- C2: WHERE filter + projection, 1e8 rows x 8 f32 columns (configs[1]): SELECT rowid, c0, c2 WHERE c1 > 0.5
    w = w_c2(torch, eng, dev, a.config_scale)
    out["C2_filter_proj"] = timed(w, warm=3, reps=10)
    out["C2_filter_proj"]["survivors"] = out["C2_filter_proj"]["result_shape"][0]
    sink()
    t8 = w["keep"][0]
    n2 = w["rows"]
    out["C1_projection"] = timed({"run": lambda: eng.query_sel(t8, [0, 2]), "bytes": lambda r: 16.0 * n2, "rows": n2,
                                  "info": {"statement": "SELECT c0,c2 FROM t8 (query_sel, main.fut:7)"}}, warm=3, reps=10)
    t8.free()
    del w, t8
    torch.cuda.empty_cache()

    # ---- the reference's own entries and ORDER BY at 1e8 rows (not BASELINE configs; the operators behind them)
    for name, wl, kw in (("REF_query_groupby_dense", "refgb", {}), ("REF_query_groupby_hash", "refgb_hash", {}), ("ORDER_BY", "sort20", {}),
                         ("ORDER_BY_32bit", "sort32", {}), ("ORDER_BY_i64", "sort64", {}), ("ORDER_BY_sorted_column", "sort20_sorted", {}), ("REF_join_u32", "join_u32", {}),
                         ("REF_join_u32_sorted_probe", "join_u32_sorted", {}), ("C4_join_share", "join_c4", {})):
        w = WORKLOADS[wl](torch, eng, dev, a.config_scale)
        out[name] = timed(w, **kw)
        if wl.startswith("refgb"):
            out[name]["groups"] = out[name]["result_shape"][0]
            out[name]["path"] = eng.last_groupby_path()
        if wl.startswith("join"):
            out[name]["path"] = eng.last_join_path()
            out[name]["pairs"] = out[name]["result_shape"][0]
            out[name]["pairs_match_the_expected_count"] = out[name]["pairs"] == out[name].get("pairs_expected")
        for t_ in w["keep"]:
            if hasattr(t_, "free"):
                t_.free()
        del w
        torch.cuda.empty_cache()
        sink()
    shape = [None]

    # ---- C5: one GPU's share of the full pipeline (configs[4]): 5e8 rows x (i32 key + 16 f32 columns), through sql()
    from harkdb_amd import FutharkContext
    n5 = int(5e8 * a.config_scale) // 4 * 4
    fc = FutharkContext.__new__(FutharkContext)
    fc.FutEnv, fc.tables, fc.sql_mode = eng, {}, True
    c5 = [torch.empty(n5, dtype=torch.float32, device=dev) for _ in range(16)]
    key = torch.empty(n5, dtype=torch.int32, device=dev)
    for jj in range(0, 16, 2):
        eng.gen_columns(SEED + jj, 0, n5, 1 << 20, False, c5[jj].data_ptr(), key.data_ptr() if jj == 0 else None, c5[jj + 1].data_ptr())
    fc.create_table_from_device("t", ["k"] + [f"c{q}" for q in range(16)], [key.data_ptr()] + [c.data_ptr() for c in c5],
                                [np.int32] + [np.float32] * 16, n5, keepalive=(key, c5))
    for name, q, ncols in (
            ("C5_pipeline_share", "select k, sum(c3), count(*), avg(c3) from t where c1 > 0.5 group by k having count(*) > 250 order by sum(c3) desc limit 10", 3),
            ("C5_three_aggregates", "select k, sum(c3), max(c7), min(c9), count(*) from t where c1 > 0.5 group by k having count(*) > 250 order by sum(c3) desc limit 10", 5)):
        res = [None]

        def c5q():
            res[0] = fc.sql(q)

        ms = event_ms(torch, c5q, warm=2, reps=5)
        # bytes the statement STREAMS (model): the key, the predicate column and the columns of the aggregates HAVING / ORDER BY
        # mention, 4 B each; with late aggregation (C5_three_aggregates) max(c7) / min(c9) are computed for the LIMIT groups only
        # by a second pass over the KEY column alone -- c7 / c9 are read for the rows of ten groups.  (Round 4 divided all five
        # referenced columns by the time: a "fraction" late aggregation could push past 1.)
        streamed = 4.0 * 3 * n5 + (4.0 * n5 if ncols == 5 else 0.0) + 16.0 * (1 << 20)
        out[name] = entry(ms, streamed, n5, referenced_columns=ncols, referenced_column_bytes=4.0 * ncols * n5, result_rows=int(res[0].shape[0]), statement=q,
                          bytes_model="k, c1, c3 streamed once" + (" + k once more by the late-aggregation pass (c7, c9: rows of the 10 surviving groups only)" if ncols == 5 else "")
                                      + "; measured bytes per kernel: profiles/r05_op_traffic.txt (c5_pipeline / c5_three)",
                          note="1/8 of configs[4] (4e9 rows x 16 f32 columns); end to end through FutharkContext.sql() incl. the LIMIT-row download")
        sink()
    out["C5_three_aggregates"]["note"] += ("; max(c7) and min(c9) are computed for the LIMIT surviving groups only (late aggregation: "
                                           "hark_entry_filter_groupby_subset, harkdb_amd/context.py; HARK_NO_LATE_AGG=1 switches it off)")
    # the same three aggregates for EVERY group, result left on the device (no HAVING / ORDER BY / LIMIT, no download):
    # what several aggregates cost in general -- ONE triple pass (sum(c3), max(c7), min(c9): 14-byte entries; round 3: a pair
    # pass + a single pass)
    dev_t = fc.tables["t"]._device

    def c5g():
        r = eng.filter_groupby(dev_t, [(2, ">", 0.5)], 0, [("sum", 4), ("max", 8), ("min", 10), ("count", 0)])
        shape[0] = r.shape
        r.free()

    ms = event_ms(torch, c5g, warm=2, reps=5)
    passes = eng.last_groupby_passes()
    out["C5_three_aggregates_all_groups"] = entry(ms, 4.0 * 5 * n5 + 24.0 * (1 << 20), n5, referenced_columns=5, result_rows=int(shape[0][0]),
                                                  statement="select k, sum(c3), max(c7), min(c9), count(*) from t where c1 > 0.5 group by k  (device result)",
                                                  passes_over_the_rows=passes,
                                                  note="hark_entry_filter_groupby: one triple pass (k_fgb_dense_multi); HARK_NO_TRIPLE_PASS=1: pair pass (sum(c3) + max(c7)) + single pass (min(c9))")
    sink()
    fc.drop_table("t")
    del c5, key
    torch.cuda.empty_cache()
    return out


def write_json_atomically(path, obj):
    tmp = path + ".tmp"
    with open(tmp, "w") as f:
        json.dump(obj, f)
    os.replace(tmp, path)


def configs_child_main(a):
    """`bench.py --configs-child FILE`: the extra configs in a process of their own (started by run_configs_child).  The
    results so far are written to FILE after every config: what a child that is killed, faults or hangs leaves behind."""
    import torch
    from harkdb_amd.engine import Engine
    from harkdb_amd import dist as hd
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    eng = Engine(0)
    hd.share_stream(eng, dev)
    out = {}
    write_json_atomically(a.configs_child, out)
    stall = os.environ.get("HARK_BENCH_CHILD_STALL")           # tests: hang (as a wedged kernel would) once this config is written

    def sink():
        write_json_atomically(a.configs_child, out)
        if stall and stall in out:
            time.sleep(1e6)

    extra_configs(torch, eng, dev, a, sink, out)
    out["complete"] = True
    write_json_atomically(a.configs_child, out)


def run_configs_child(a, budget_s):
    """Start `bench.py --configs-child` exactly as the --pmc children are started -- a NEW process (this one has initialised
    the GPU: never an exec), its own process group, a wall-clock budget -- and return its configs.  Whatever happens to the
    child (non-zero exit, a GPU fault, a hang killed at the budget) this returns a dict: the configs it finished plus "error"."""
    import subprocess
    import tempfile
    tmp = tempfile.mkdtemp(prefix="hark_cfg_", dir="/tmp")
    path = os.path.join(tmp, "configs.json")
    cmd = [sys.executable, os.path.abspath(__file__), "--configs-child", path, "--rows", str(int(a.rows)), "--groups", str(int(a.groups)),
           "--exact", str(a.exact), "--config-scale", str(a.config_scale)]
    env = {kk: vv for kk, vv in os.environ.items() if kk not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "HARK_FORCE_PIPELINE")}
    err_note = None
    t0 = time.perf_counter()
    try:
        pr = subprocess.Popen(cmd, env=env, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, start_new_session=True)
        CHILD_PGIDS.append(pr.pid)
        try:
            _, err = pr.communicate(timeout=budget_s)
            if pr.returncode:
                err_note = f"configs child exited with rc {pr.returncode}: {err.decode(errors='replace')[-400:]}"
        except subprocess.TimeoutExpired:
            os.killpg(pr.pid, 9)                                  # the exact process group started here
            pr.wait()
            err_note = f"configs child killed at its wall-clock budget of {budget_s:.0f} s"
        finally:
            CHILD_PGIDS.remove(pr.pid)
    except Exception as e:
        err_note = "configs child could not be started: " + repr(e)
    out = {}
    try:
        out = json.load(open(path))
    except Exception as e:
        if err_note is None:
            err_note = "configs child left no readable result: " + repr(e)
    import shutil
    shutil.rmtree(tmp, ignore_errors=True)
    if err_note is None and not out.pop("complete", False):
        err_note = "configs child ended without completing"
    out.pop("complete", None)
    if err_note:
        out["error"] = err_note
    out["child_wall_s"] = time.perf_counter() - t0
    return out


def main():
    a = parse_args()
    if a.configs_child:
        return configs_child_main(a)
    if a.gpus > 1 and "RANK" not in os.environ:
        launch_ranks(a)
    if a.stub:
        return stub_main(a)
    import signal
    import torch
    import torch.distributed as dist
    from harkdb_amd import dist as hd
    from harkdb_amd.engine import Engine, FgbPlan

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    rank, local, world = hd.init_process_group("cuda")
    if world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    G = int(a.groups)

    eng = Engine(local)
    stream = hd.share_stream(eng, dev)                          # kernels, events and collectives share one (non-default) stream
    knobs = {"timing": 1}
    if a.algo:
        knobs["algo"] = a.algo
    if a.chunk_rows:
        knobs["chunk_rows"] = a.chunk_rows
    # With peers, the producer may leave a few CUs to RCCL: its 256 workgroups of 1024 threads otherwise hold every CU for the
    # whole pass, and the all-reduce of the previous step (pipelined mode) could only start in the producer's tail.  How
    # many (and whether pipelining two plans pays at all) depends on hardware this code was never run on with peers, so it is
    # MEASURED at start-up of every mode, before the warm-up steps: a few steps of each candidate (producer geometry x
    # pipelining x the collective), the slowest rank's time decides, every rank takes the same choice; the default candidate
    # stays unless another beats it by more than TUNE_MARGIN.  HARK_PRODUCER_WGS (0 = all CUs) / HARK_OVERLAP=0|1 /
    # HARK_ALLREDUCE each pin their own dimension only.
    multi = world > 1 or bool(os.environ.get("HARK_FORCE_PIPELINE"))

    def sync_all():
        torch.cuda.synchronize()
        if dist.is_initialized():
            dist.barrier()
        torch.cuda.synchronize()

    primed = [False]

    def run_mode(label, N, first_row, rows_total, keep):
        """One mode's columns, start-up measurement, W warm-up steps and K timed steps; a record of plain numbers (and, with
        keep, the device data the N = 1 extras go on with)."""
        p = torch.empty(N, dtype=torch.float32, device=dev)
        k = torch.empty(N, dtype=torch.int32, device=dev)
        v = torch.empty(N, dtype=torch.float32, device=dev)
        eng.gen_columns(SEED, first_row, N, G, bool(a.exact), p.data_ptr(), k.data_ptr(), v.data_ptr())
        sum_out = torch.empty(G, dtype=torch.float32, device=dev)
        cnt_out = torch.empty(G, dtype=torch.int64, device=dev)

        def make_job(wgs, overlap, how):
            kn = dict(knobs)
            if wgs > 0:
                kn["grid"] = wgs
            pl = FgbPlan(eng, N, G, **kn)
            pl2 = FgbPlan(eng, N, G, **kn) if overlap else None
            return pl, pl2, hd.ShardedFgb(eng, pl, dev, plan2=pl2, how=how)

        tuned, margin = None, None
        if multi:
            cands = tuning_candidates(os.environ)
        else:
            cands = [(int(os.environ.get("HARK_PRODUCER_WGS", "0")), False, os.environ.get("HARK_ALLREDUCE", "allreduce"))]
        if len(cands) > 1:
            tuned = {}
            for wgs, overlap, how in cands:
                pl, pl2, jb = make_job(wgs, overlap, how)
                for timed_steps in (2, 6):                        # 2 untimed, then 6 timed
                    sync_all()
                    t0 = time.perf_counter()
                    for _ in range(timed_steps):
                        jb.step(p.data_ptr(), ">", 0.5, k.data_ptr(), v.data_ptr(), N, sum_out.data_ptr(), cnt_out.data_ptr(), check=False)
                    jb.flush()
                    sync_all()
                    dt = time.perf_counter() - t0
                tt = torch.tensor([dt / 6 * 1e3], dtype=torch.float64, device=dev)
                if dist.is_initialized():
                    dist.all_reduce(tt, op=dist.ReduceOp.MAX)     # the same number on every rank: the same choice on every rank
                tuned[candidate_name(wgs, overlap, how)] = float(tt.item())
                del jb
                pl.free()
                if pl2 is not None:
                    pl2.free()
            chosen, margin = pick_candidate(tuned)
            producer_wgs, overlap, how = cands[list(tuned).index(chosen)]
        else:
            producer_wgs, overlap, how = cands[0]
        # N > 1: a second plan lets step i's all-reduce run on RCCL's stream beside the kernels of step i+1 (dist.ShardedFgb)
        # (HARK_FORCE_PIPELINE=1: also with one rank -- tests/test_gpu_bench_rank.py runs the pipelined, asynchronous
        # all-reduce path and the start-up measurement under RCCL on the one GPU a test box has)
        plan, plan2, job = make_job(producer_wgs, overlap, how)

        def step():
            # check=False: no host round trip inside the loop; job.flush() reads the sticky error words after the last step
            job.step(p.data_ptr(), ">", 0.5, k.data_ptr(), v.data_ptr(), N, sum_out.data_ptr(), cnt_out.data_ptr(), check=False)

        # Setup, not warm-up: the path's kernels are loaded (code objects page in on first launch -- 0.3 s on a fresh box) by ONE
        # pass over the first 65536 rows with a small plan of its own; the workload's plans, slabs and rows are not touched.
        # (not in a --pmc-child and not under a profiler: there every launch of the path's kernels is counted or averaged -- the
        # rocprofv3 kernel statistics committed under profiles/ must show headline launches only, so that their average duration
        # is the one this line reports)
        if not primed[0] and not a.pmc_child and not under_profiler():
            prime_n = min(N, 1 << 16)
            prime = FgbPlan(eng, prime_n, G, **knobs)
            prime.run(p.data_ptr(), ">", 0.5, k.data_ptr(), v.data_ptr(), prime_n)
            prime.finish(sum_out.data_ptr(), cnt_out.data_ptr())
            torch.cuda.synchronize()
            prime.free()
        primed[0] = True
        for _ in range(a.warmup):
            step()
        job.flush()
        plan.timing()                                             # drop warm-up events
        if plan2 is not None:
            plan2.timing()
        sync_all()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            step()
        job.flush()                                               # the last step's all-reduce + finish belong to the timed region
        sync_all()
        elapsed_local = time.perf_counter() - t0
        rec = {"mode": label, "rows_local": N, "first_row": first_row, "rows_total": rows_total, "elapsed_local": elapsed_local}
        if a.pmc_child:                                           # under rocprofv3 --pmc: the counters are all that is wanted
            return rec
        t = torch.tensor([elapsed_local], dtype=torch.float64, device=dev)
        per_rank = [elapsed_local]
        if dist.is_initialized():
            gathered = [torch.zeros_like(t) for _ in range(world)]
            dist.all_gather(gathered, t)
            per_rank = [float(x.item()) for x in gathered]
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        ms_by_kind, launches = plan.timing()
        if plan2 is not None:                                     # the steps alternated between the two plans
            ms2, l2 = plan2.timing()
            ms_by_kind = {kk: ms_by_kind[kk] + ms2[kk] for kk in ms_by_kind}
            launches = {kk: launches[kk] + l2[kk] for kk in launches}
        # ---- size-independent checks on the merged full-size result
        survivors_local = int((p > 0.5).sum().item())
        st = torch.tensor([survivors_local, N], dtype=torch.int64, device=dev)
        if dist.is_initialized():
            dist.all_reduce(st)
        check = {"count_checksum": int(cnt_out.sum().item()) == int(st[0].item())}
        if a.exact:                                               # integer-valued v: the f64 checksum of sums is exact
            sv = torch.where(p > 0.5, v, torch.zeros_like(v)).to(torch.float64).sum()
            if dist.is_initialized():
                dist.all_reduce(sv)
            check["sum_checksum"] = float(sum_out.to(torch.float64).sum().item()) == float(sv.item())
        if world > 1:
            check["rows_seen_by_all_ranks"] = int(st[1].item())
            check["all_rows_seen"] = int(st[1].item()) == rows_total
        rec.update({"elapsed": float(t.item()), "per_rank": per_rank, "ms_by_kind": ms_by_kind, "launches": launches, "tuned": tuned,
                    "margin": margin, "producer_wgs": producer_wgs, "overlap": bool(overlap), "how": how, "pipelined": plan2 is not None,
                    "check": check, "survivors": int(st[0].item())})
        if keep:
            rec["data"] = (p, k, v, sum_out, cnt_out, plan, plan2, job)
        else:
            del job
            plan.free()
            if plan2 is not None:
                plan2.free()
            del p, k, v, sum_out, cnt_out
            torch.cuda.empty_cache()
        return rec

    def mode_record(rec):
        """A mode's numbers as they appear in the line: whole-job rows/s, ms per step, per-GPU roofline fraction of the path."""
        N = rec["rows_local"]
        ms_step = rec["elapsed"] / a.steps * 1e3
        ms_by_kind, launches = rec["ms_by_kind"], rec["launches"]
        dom = max(ms_by_kind, key=lambda kk: ms_by_kind[kk])
        dom_launches = max(1, launches[dom])
        dom_ms = ms_by_kind[dom] / dom_launches
        rows_per_launch = N * a.steps / dom_launches              # a chunked producer sees chunk_rows per launch
        alg_bytes = 12.0 * rows_per_launch + (16.0 * G if dom != "producer" else 0.0)
        dom_achieved = alg_bytes / (dom_ms * 1e-3) / 1e9
        kernel_name = {"single": "fgb_lds_kernel", "producer": "fgb_part_kernel", "consumer": "fgb_agg6_kernel"}[dom]
        path_bytes = 12.0 * N + 16.0 * G                          # per GPU: every rank moves its own 12 B/row in the same wall time
        kernels_ms = sum(ms_by_kind.values()) / a.steps
        path_achieved = path_bytes / (ms_step * 1e-3) / 1e9
        with_peers = dist.is_initialized()
        return {
            "value": rec["rows_total"] / (rec["elapsed"] / a.steps), "unit": "rows/s", "ms_per_step": ms_step,
            "scaling": "weak" if rec["mode"] == "weak" else "strong",
            "ms_per_step_by_rank": [x / a.steps * 1e3 for x in rec["per_rank"]],
            "config": {"workload": "BASELINE configs[2] + filter: SELECT k,SUM(v),COUNT(*) FROM t WHERE p>0.5 GROUP BY k",
                       "mode": rec["mode"], "rows_total": rec["rows_total"], "rows_per_gpu": rec["rows_total"] // world, "rows_this_rank": N,
                       "first_row_of_rank0": rec["first_row"], "groups": G, "selectivity": 0.5, "columns": "p f32, k i32, v f32 (HBM-resident)",
                       "exact_values": bool(a.exact),
                       "merge": ("RCCL " + ("reduce-scatter + all-gather" if rec["how"] == "rs_ag" else "all-reduce")
                                 + " of f64 sums + i64 counts" + (", overlapped with the next step's kernels" if rec["pipelined"] else "")) if with_peers else "none",
                       "pipelined_steps": rec["pipelined"], "producer_workgroups": rec["producer_wgs"] or "all CUs",
                       "allreduce": rec["how"], "overlap": rec["overlap"],
                       "measured_at_startup_ms_per_step": rec["tuned"], "startup_choice_margin_over_default": rec["margin"],
                       "startup_rule": f"the first (default) candidate unless another is more than {TUNE_MARGIN:.0%} faster"},
            "roofline": {"bound": "hbm", "kernel": kernel_name, "achieved": path_achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": path_achieved / HBM_PEAK_GBS,
                         "frac_is": "whole path per GPU: (12 B/row x rows of the rank + 16 B x groups) / wall time of a step / 8 TB/s",
                         "dominant_kernel_frac": dom_achieved / HBM_PEAK_GBS, "dominant_kernel_achieved": dom_achieved,
                         "avg_launch_ms": dom_ms, "launches": dom_launches, "algorithmic_bytes_per_launch": alg_bytes,
                         "algorithmic_bytes_per_step": path_bytes},
            "hot_path": {"kernel_ms_per_step": kernels_ms, "by_kernel_ms_per_step": {kk: ms_by_kind[kk] / a.steps for kk in ms_by_kind},
                         "algorithmic_GBps_per_gpu": path_bytes / (kernels_ms * 1e-3) / 1e9,
                         "frac_of_peak_all_kernels": path_bytes / (kernels_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                         "frac_of_peak_wall": path_achieved / HBM_PEAK_GBS,
                         "step_minus_kernels_ms": ms_step - kernels_ms},
            "check": rec["check"],
        }

    specs = mode_specs(a.rows, rank, world, a.modes, a.first_row)
    recs = []
    for j, (label, n_local, first_row, rows_total) in enumerate(specs):
        recs.append(run_mode(label, n_local, first_row, rows_total, keep=(world == 1 and j == 0)))
        if a.pmc_child:
            print(json.dumps({"pmc_child": True, "ms_per_step": recs[0]["elapsed_local"] / a.steps * 1e3}), flush=True)
            return
    head = recs[0]
    N = head["rows_local"]

    state = {"out": None, "printed": False}

    def summarise(line):
        """LAST key of the line: {config: [ms, fraction of the HBM peak]} for the headline and every entry of `configs` -- the driver keeps
        the tail of stdout, and the configs that sit at the front of the object (C3_no_filter, G16 / G4096 / G13000, the sweeps) were
        cut off there.  Nothing is printed behind it."""
        summ = {"headline": [round(line["ms_per_step"], 3), round(line.get("roofline", {}).get("frac", 0.0) or 0.0, 4)]}

        def walk(prefix, d, depth):
            for name, e in d.items():
                if not isinstance(e, dict):
                    continue
                if isinstance(e.get("ms"), (int, float)):
                    f = e.get("frac_of_peak")
                    summ[prefix + name] = [round(e["ms"], 3), round(f, 4) if isinstance(f, (int, float)) else None]
                elif depth < 2:
                    walk(prefix + name + ".", e, depth + 1)
        if isinstance(line.get("configs"), dict):
            walk("", line["configs"], 0)
        line.pop("summary", None)
        line["summary"] = summ

    def emit():
        if rank == 0 and state["out"] is not None and not state["printed"]:
            state["printed"] = True
            try:
                summarise(state["out"])
            except Exception as e:
                state["out"]["summary"] = {"error": repr(e)}
            for _ in range(20):                                   # the watchdog thread may serialise while the main thread adds keys
                try:
                    text = json.dumps(state["out"])
                    break
                except RuntimeError:
                    time.sleep(0.05)
            else:
                text = json.dumps({kk: state["out"][kk] for kk in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                                                                   "scaling", "vs_baseline", "dtype", "data")})
            print(text, flush=True)

    if rank == 0:
        out = mode_record(head)
        line = {"metric": "rows/sec, 1B-row f32 filter->group-by (SUM,COUNT), 2^20 groups", "value": out["value"], "unit": "rows/s",
                "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": out["ms_per_step"],
                "higher_is_better": True, "scaling": out["scaling"], "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                "rccl_ranks": dist.get_world_size() if dist.is_initialized() else 1,
                "library_linked_from_the_sources_in_the_tree": library_matches_sources() if not os.environ.get("HARK_LIB") else None,
                "scaling_note": "the metric's own configuration: ONE rows_total-row table at every N (strong scaling); at N = 1 strong and weak "
                                "coincide; with N > 1 the weak-scaling run of the same invocation is the `weak` sub-record",
                "ms_per_step_by_rank": out["ms_per_step_by_rank"], "config": out["config"], "roofline": out["roofline"],
                "hot_path": out["hot_path"], "check": out["check"], "weak": None, "cpu_baseline": None}
        for r in recs[1:]:
            line[r["mode"]] = mode_record(r)
        if world > 1 and len(recs) > 1 and recs[1]["mode"] == "weak":
            # what the driver's scaling efficiency will be computed from (value_N / (N x value_1)) needs the N = 1 line; what CAN be
            # said inside one run: the strong step against 1/N of the weak step (equal if the kernels scaled linearly in rows
            # and the merge were free)
            line["strong_step_over_weak_step_divided_by_N"] = line["ms_per_step"] / (line["weak"]["ms_per_step"] / world)
        state["out"] = line

        def give_up(why):                                         # the headline is in hand -- print it, whatever the extras are doing
            line.setdefault("interrupted", why + " while the extras (probes / checks / PMC children / configs child / CPU port) ran")
            emit()
            sys.stdout.flush()
            for pg in list(CHILD_PGIDS):                          # children started here (exact process groups), not patterns
                try:
                    os.killpg(pg, 9)
                except OSError:
                    pass
            os._exit(0)

        signal.signal(signal.SIGTERM, lambda signum, frame: give_up(f"signal {signum}"))
        if world == 1 and a.extras_budget > 0:
            import threading
            # a hang inside a blocking GPU call never returns to the interpreter, so no signal handler would run: a daemon
            # thread does (blocking torch / ctypes calls release the GIL)
            wd = threading.Timer(a.extras_budget, lambda: give_up(f"the extras' wall-clock budget of {a.extras_budget:.0f} s ran out"))
            wd.daemon = True
            wd.start()

    if world == 1 and rank == 0:
        single_gpu_extras(a, torch, eng, dev, head, state["out"], G)
    emit()
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


def single_gpu_extras(a, torch, eng, dev, head, out, G):
    """N = 1, after the timed region, every part bounded and none able to lose the headline (out is complete as it stands):
    stream probes, the tolerance variant, the HIP result on the CPU sample's rows, then -- tables freed -- the --pmc children,
    the configs child, the CPU port."""
    from harkdb_amd.engine import FgbPlan
    N = head["rows_local"]
    p, k, v, sum_out, cnt_out, plan, plan2, job = head.pop("data")
    ms_step = out["ms_per_step"]
    path_bytes = 12.0 * N + 16.0 * G
    path_achieved = out["roofline"]["achieved"]
    dom_launches = out["roofline"]["launches"]
    # ---- what plain streams with the path's byte mix reach on THIS device, in this process (the practical ceilings):
    #      [1] read the three columns once; [2] read them and write 3 B/row (the producer's mix at 50 % selectivity,
    #      6-byte pairs); [3] read those 3 B/row back (the consumer).  [2] + [3] = what plain streams with the byte mix
    #      of a two-pass design take on this box (a reference, not a bound).
    try:
        nb = (N * 4) & ~255
        fold = torch.zeros(1, dtype=torch.int64, device=dev)
        scratch = torch.empty(nb * 12 // 16 + 4096, dtype=torch.uint8, device=dev)
        cols3 = [p.data_ptr(), k.data_ptr(), v.data_ptr()]

        def ev(fn):
            return event_ms(torch, fn, warm=1, reps=5)

        read_ms = ev(lambda: eng.stream_read(cols3, nb, fold.data_ptr()))
        mix_ms = ev(lambda: eng.stream_mix(cols3, nb, scratch.data_ptr(), 12))
        back_bytes = (nb * 12 // 16) & ~15
        back_ms = ev(lambda: eng.stream_read([scratch.data_ptr()], back_bytes, fold.data_ptr()))
        stream_gbs = 3 * nb / (read_ms * 1e-3) / 1e9
        del scratch
        floor_ms = mix_ms + back_ms
        out["roofline"].update({
            "measured_stream_read": stream_gbs, "frac_of_measured_stream_read": path_achieved / stream_gbs,
            "probes_ms": {"read_3_columns": read_ms, "read_3_columns_write_3B_per_row": mix_ms, "read_back_3B_per_row": back_ms},
            "two_pass_probe_ms": floor_ms, "two_pass_probe_frac_of_peak": path_bytes / (floor_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
            "step_over_two_pass_probe": ms_step / floor_ms,
            "two_pass_probe_is": "probes [2] + [3]: plain streaming kernels moving the byte mix of a write-once / read-once partition design "
                                 "(a reference for THIS box, not a bound: on the faster boxes the partition kernel itself runs below probe [2])",
            "why_two_passes": "2^20 groups x 12 B = 12 MiB of accumulators fit no LDS (160 KiB/CU) and no XCD L2 (4 MiB); "
                              "scattered global atomics retire ~25 G/s (DESIGN.md 3.1)"})
    except Exception as e:
        out["roofline"]["probes_error"] = repr(e)

    # ---- the HIP path on exactly the rows the CPU port is timed on below (the first cpu_rows rows of this table)
    hip_sample = None
    if a.cpu_rows > 0:
        ns = min(int(a.cpu_rows), N)
        so_s, co_s = torch.empty(G, dtype=torch.float32, device=dev), torch.empty(G, dtype=torch.int64, device=dev)
        plan_s = FgbPlan(eng, ns, G)
        plan_s.run(p.data_ptr(), ">", 0.5, k.data_ptr(), v.data_ptr(), ns)
        plan_s.finish(so_s.data_ptr(), co_s.data_ptr())
        hip_sample = (ns, so_s.cpu().numpy(), co_s.cpu().numpy())
        plan_s.free()
        del so_s, co_s

    # ---- the TOLERANCE variant of the value column (v uniform in [0, 1), the 1e-5 bar of north_star), one untimed
    #      pass of the same plan at FULL size, held to a torch f64 scatter-add of the same rows; and on the CPU sample's first
    #      rows for the comparison with the port further down
    tol_sample = None
    if a.tolerance_check:
        try:
            eng.gen_columns(SEED, 0, N, G, not a.exact, None, None, v.data_ptr())   # the OTHER variant of v than the timed run's
            plan.reset()
            plan.run(p.data_ptr(), ">", 0.5, k.data_ptr(), v.data_ptr(), N)
            plan.finish(sum_out.data_ptr(), cnt_out.data_ptr())
            ref = torch.zeros(G, dtype=torch.float64, device=dev)
            for lo in range(0, N, 1 << 27):
                pp, kk_, vv = p[lo:lo + (1 << 27)], k[lo:lo + (1 << 27)], v[lo:lo + (1 << 27)]
                keep_ = pp > 0.5
                ref.index_add_(0, kk_[keep_].to(torch.int64), vv[keep_].to(torch.float64))
                del pp, kk_, vv, keep_
            got = sum_out.to(torch.float64)
            rel = ((got - ref).abs() / ref.abs().clamp_min(1e-30)).max().item()
            out["check"]["tolerance_variant"] = {
                "values": "integer-valued 0..15" if not a.exact else "uniform [0,1) = ((h>>40)&0xFFFFFF)/2^24", "rows": N, "groups": G,
                "max_relative_error_vs_f64_scatter_add": rel, "bar": 1e-5, "within_bar": bool(rel <= 1e-5),
                "reference": "torch f64 index_add_ over the same device columns (a second implementation at full size, not the oracle)"}
            if a.cpu_rows > 0:
                nt = min(int(a.cpu_rows), N, 50_000_000)
                so_t, co_t = torch.empty(G, dtype=torch.float32, device=dev), torch.empty(G, dtype=torch.int64, device=dev)
                plan_t = FgbPlan(eng, nt, G)
                plan_t.run(p.data_ptr(), ">", 0.5, k.data_ptr(), v.data_ptr(), nt)
                plan_t.finish(so_t.data_ptr(), co_t.data_ptr())
                tol_sample = (nt, so_t.cpu().numpy(), co_t.cpu().numpy())
                plan_t.free()
                del so_t, co_t
            del ref, got
        except Exception as e:
            out["check"]["tolerance_variant"] = {"error": repr(e)}

    # ---- free this process's tables before any child runs
    del job
    plan.free()
    if plan2 is not None:
        plan2.free()
    del p, k, v, sum_out, cnt_out
    torch.cuda.empty_cache()
    try:
        eng.lib.hark_context_trim(eng.ctx)
    except Exception:
        pass

    # ---- HBM traffic of the path's kernels: measured in THIS run by two rocprofv3 --pmc child runs of the headline steps
    #      (measure_traffic), else read from the newest committed PMC passes of the same workload (marked as such)
    traffic, traffic_src, traffic_live, traffic_detail, traffic_note = None, None, False, None, None
    want_pmc = a.pmc == 1 or (a.pmc < 0 and not a.algo and not a.chunk_rows)
    if want_pmc:
        traffic_detail, traffic_note = measure_traffic(N, G)
        if traffic_detail:
            traffic = sum(kk["hbm_bytes_per_launch_corrected"] for kk in traffic_detail.values()) * (dom_launches / a.steps)
            traffic_src, traffic_live = f"rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE child runs of this bench.py ({build_id()})", True
        elif a.pmc == 1:
            out["roofline"]["traffic_note"] = traffic_note
            print(json.dumps(out), flush=True)
            raise SystemExit("--pmc 1: " + str(traffic_note))
    if traffic is None:
        for cand in ("r06_pmc_fgb.json", "r05_pmc_fgb.json", "r04_pmc_fgb.json", "r03_pmc_fgb.json", "r02_pmc_fgb.json", "r01_pmc_fgb.json"):
            try:
                pmc = json.load(open(os.path.join(ROOT, "profiles", cand)))
                if (pmc["config"]["rows_per_gpu"] == N and pmc["config"]["groups"] == G and not a.algo and not a.chunk_rows
                        and dom_launches == a.steps * pmc["config"]["producer_launches_per_step"]):
                    traffic = sum(kk["hbm_bytes_per_launch_corrected"] for kk in pmc["kernels"].values())
                    traffic_src = "profiles/" + cand + (f" (git {pmc['git']})" if pmc.get("git") else "")
                    break
            except Exception:
                pass
    out["roofline"].update({
        "traffic": traffic, "traffic_source": traffic_src, "traffic_measured_in_run": traffic_live,
        "traffic_is": "HBM bytes per step of the path's kernels (producer + consumer): 2 x FETCH_SIZE + WRITE_SIZE, KiB -> bytes",
        "traffic_by_kernel": traffic_detail, "traffic_note": traffic_note,
        "traffic_over_algorithmic": (traffic / path_bytes) if traffic else None,
        # what the HBM actually moves per second on this path (PMC bytes over this run's step time)
        "traffic_GBps": (traffic / (ms_step * 1e-3) / 1e9) if traffic else None,
        "traffic_frac_of_peak": (traffic / (ms_step * 1e-3) / 1e9 / HBM_PEAK_GBS) if traffic else None})

    if a.configs:
        out["configs"] = run_configs_child(a, a.configs_budget)

    if hip_sample is not None:
        try:
            ns, hs, hc = hip_sample
            base, (bk, bs, bc) = cpu_baseline(ns, G, a.exact)
            out["cpu_baseline"] = base
            out["cpu_baseline"]["gpu_speedup_same_metric"] = out["value"] / base["value"]
            # same rows, both implementations: every group the port reports, and no group besides them
            idx = bk.astype(np.int64)
            same_counts = bool(np.array_equal(hc[idx], bc.astype(np.int64))) and int(hc.sum()) == int(bc.sum())
            if a.exact:
                same_sums = bool(np.array_equal(hs[idx], bs.astype(np.float32)))
            else:                                                 # 1e-5 relative (BASELINE.json north_star), sums of non-negative values
                same_sums = bool(np.all(np.abs(hs[idx].astype(np.float64) - bs.astype(np.float64)) <= 1e-5 * np.abs(bs.astype(np.float64)) + 1e-30))
            out["check"]["hip_equals_cpu_port_on_sample"] = same_counts and same_sums
            out["check"]["hip_vs_cpu_port"] = {"rows": ns, "groups_compared": int(len(bk)), "counts_equal": same_counts, "sums_equal": same_sums,
                                               "sums_bar": "bit-exact (integer-valued f32)" if a.exact else "1e-5 relative"}
            if tol_sample is not None and isinstance(out["check"].get("tolerance_variant"), dict):
                from oracle import oracle as ora
                nt, ts_, tc_ = tol_sample
                hp, hk, hv = ora.gen_columns(SEED, 0, nt, G, not a.exact)
                t0 = time.perf_counter()
                tk, tsum, tcnt = ora.filter_groupby_refalgo_f32(hp, hk, hv, ">", 0.5)
                tdt = time.perf_counter() - t0
                ti = tk.astype(np.int64)
                d = np.abs(ts_[ti].astype(np.float64) - tsum.astype(np.float64)) / np.maximum(np.abs(tsum.astype(np.float64)), 1e-30)
                out["check"]["tolerance_variant"]["hip_vs_cpu_port"] = {
                    "rows": nt, "groups_compared": int(len(tk)), "counts_equal": bool(np.array_equal(tc_[ti], tcnt.astype(np.int64)) and int(tc_.sum()) == int(tcnt.sum())),
                    "max_relative_error": float(d.max()) if len(d) else 0.0, "bar": 1e-5, "within_bar": bool((d <= 1e-5).all()),
                    "port_seconds": tdt, "port": "oracle/hark_oracle.c ora_filter_groupby_refalgo_f32 (sequential f32 fold in table order)"}
        except Exception as e:
            out["cpu_baseline_error"] = repr(e)


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Headline benchmark of BASELINE.json on MI355X.

    python bench.py --gpus N --steps K --warmup W

One step = one pass of the hot path over one batch of synthetic input:
    SELECT k, SUM(v), COUNT(*) FROM t WHERE p > 0.5 GROUP BY k
over three device-resident columns (p f32, k i32, v f32) of 1e9 rows per GPU
with 2^20 groups (BASELINE.json configs[2] with the filter).
Weak scaling: rank r holds rows [r*1e9, (r+1)*1e9) of an N*1e9-row table, runs
the fused HIP kernels on its shard, the per-GPU partial aggregates (16 B x 2^20)
are summed with an RCCL all-reduce, and the merged table is finalised.

Prints ONE JSON line on rank 0 (contract in the task statement) carrying
`roofline` (dominant kernel, HIP-event timed on the launch stream) and
`cpu_baseline` (the oracle's port of the reference's 32-pass algorithm timed
on one host core over a bounded sample; rank 0, N=1 only).
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
HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300-6500 GB/s is what a stream reaches


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--rows", type=float, default=1e9, help="rows per GPU")
    ap.add_argument("--groups", type=int, default=1 << 20)
    ap.add_argument("--cpu-rows", type=float, default=1.5e8, help="rows of the CPU baseline sample (0 disables)")
    ap.add_argument("--exact", type=int, default=1, help="1: integer-valued v (bit-exact check), 0: uniform [0,1)")
    ap.add_argument("--algo", type=int, default=0)
    ap.add_argument("--chunk-rows", type=int, default=0)
    return ap.parse_args()


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


def main():
    a = parse_args()
    if a.gpus > 1 and "RANK" not in os.environ:
        launch_ranks(a)
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
    N, G = int(a.rows), int(a.groups)

    eng = Engine(local)
    stream = hd.share_stream(eng, dev)                          # kernels, events and collectives share one (non-default) stream
    p = torch.empty(N, dtype=torch.float32, device=dev)
    k = torch.empty(N, dtype=torch.int32, device=dev)
    v = torch.empty(N, dtype=torch.float32, device=dev)
    eng.gen_columns(SEED, rank * N, N, G, bool(a.exact), p.data_ptr(), k.data_ptr(), v.data_ptr())
    sum_out = torch.empty(G, dtype=torch.float32, device=dev)
    cnt_out = torch.empty(G, dtype=torch.int64, device=dev)
    knobs = {"timing": 1}
    if a.algo:
        knobs["algo"] = a.algo
    if a.chunk_rows:
        knobs["chunk_rows"] = a.chunk_rows
    plan = FgbPlan(eng, N, G, **knobs)
    job = hd.ShardedFgb(eng, plan, dev)

    def step():
        job.step(p.data_ptr(), ">", 0.5, k.data_ptr(), v.data_ptr(), N, sum_out.data_ptr(), cnt_out.data_ptr())

    def fence():
        torch.cuda.synchronize()
        if dist.is_initialized():
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(a.warmup):
        step()
    plan.timing()                                             # drop warm-up events
    fence()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if dist.is_initialized():
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    ms_by_kind, launches = plan.timing()

    # ---- what a pure read stream of the same three columns reaches on THIS device (the practical ceiling)
    fold = torch.zeros(1, dtype=torch.int64, device=dev)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

    def read_all():                                              # the three columns at once, like the fused kernels read them
        eng.stream_read([p.data_ptr(), k.data_ptr(), v.data_ptr()], (N * 4) & ~15, fold.data_ptr())

    read_all()
    torch.cuda.synchronize()
    e0.record()
    for _ in range(5):
        read_all()
    e1.record()
    torch.cuda.synchronize()
    stream_gbs = 5 * 3 * ((N * 4) & ~15) / (e0.elapsed_time(e1) * 1e-3) / 1e9

    # ---- size-independent checks on the full-size result (rank-local where possible)
    survivors_local = int((p > 0.5).sum().item())
    st = torch.tensor([survivors_local], dtype=torch.int64, device=dev)
    if dist.is_initialized():
        dist.all_reduce(st)
    total_cnt = int(cnt_out.sum().item())
    check = {"count_checksum": total_cnt == int(st.item())}
    if a.exact:                                               # integer-valued v: the f64 checksum of sums is exact
        sv = torch.where(p > 0.5, v, torch.zeros_like(v)).to(torch.float64).sum()
        if dist.is_initialized():
            dist.all_reduce(sv)
        check["sum_checksum"] = float(sum_out.to(torch.float64).sum().item()) == float(sv.item())

    if rank == 0:
        rows_total = N * world
        ms_step = elapsed / a.steps * 1e3
        dom = max(ms_by_kind, key=lambda kk: ms_by_kind[kk])
        dom_launches = max(1, launches[dom])
        dom_ms = ms_by_kind[dom] / dom_launches
        rows_per_launch = N * a.steps / dom_launches          # a chunked producer sees chunk_rows per launch
        alg_bytes = 12.0 * rows_per_launch + (16.0 * G if dom != "producer" else 0.0)
        achieved = alg_bytes / (dom_ms * 1e-3) / 1e9
        kernel_name = {"single": "fgb_lds_kernel", "producer": "fgb_part_kernel", "consumer": "fgb_agg6_kernel"}[dom]
        # HBM traffic of the dominant kernel from the committed PMC passes of this same workload
        # (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate runs, FETCH_SIZE doubled per the gfx950 note)
        traffic, traffic_src = None, None
        try:
            pmc = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc_fgb.json")))
            if (pmc["config"]["rows_per_gpu"] == N and pmc["config"]["groups"] == G and not a.algo and not a.chunk_rows
                    and dom_launches == a.steps * pmc["config"]["producer_launches_per_step"]):
                traffic = pmc["kernels"][kernel_name]["hbm_bytes_per_launch_corrected"]
                traffic_src = "profiles/r01_pmc_fgb.json"
        except Exception:
            pass
        path_bytes = 12.0 * N + 16.0 * G
        kernels_ms = sum(ms_by_kind.values()) / a.steps
        out = {
            "metric": "rows/sec, 1B-row f32 filter->group-by (SUM,COUNT), 2^20 groups",
            "value": rows_total / (elapsed / a.steps), "unit": "rows/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": ms_step,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "BASELINE configs[2] + filter: SELECT k,SUM(v),COUNT(*) FROM t WHERE p>0.5 GROUP BY k",
                       "rows_per_gpu": N, "groups": G, "selectivity": 0.5, "columns": "p f32, k i32, v f32 (HBM-resident)",
                       "exact_values": bool(a.exact), "merge": "RCCL all-reduce of f64 sums + i64 counts" if world > 1 else "none"},
            "roofline": {"bound": "hbm", "kernel": kernel_name, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                         "avg_launch_ms": dom_ms, "launches": dom_launches, "algorithmic_bytes_per_launch": alg_bytes,
                         "measured_stream_read": stream_gbs, "frac_of_measured_stream_read": achieved / stream_gbs},
            "hot_path": {"kernel_ms_per_step": kernels_ms, "by_kernel_ms_per_step": {kk: ms_by_kind[kk] / a.steps for kk in ms_by_kind},
                         "algorithmic_GBps_per_gpu": path_bytes / (kernels_ms * 1e-3) / 1e9,
                         "frac_of_peak_all_kernels": path_bytes / (kernels_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                         "frac_of_peak_wall": path_bytes / (ms_step * 1e-3) / 1e9 / HBM_PEAK_GBS},
            "check": check,
        }
        if world == 1 and a.cpu_rows > 0:
            base, (bk, bs, bc) = cpu_baseline(a.cpu_rows, G, a.exact)
            out["cpu_baseline"] = base
            out["cpu_baseline"]["gpu_speedup_same_metric"] = out["value"] / base["value"]
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out), flush=True)
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

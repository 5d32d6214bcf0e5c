"""bench.py as ONE RCCL rank on the GPU (the torchrun environment the driver sets for N > 1, with WORLD_SIZE = 1): process
group over nccl (= RCCL), the fused kernels on the shared stream, and -- with HARK_FORCE_PIPELINE -- the two-plan pipelined
step whose all-reduces are asynchronous RCCL work handles awaited one step later (harkdb_amd/dist.py ShardedFgb), i.e. the
exact code path of the N = 2, 4, 8 runs minus the peers.  The line's checksums compare the merged result with torch."""
import json
import os
import socket
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("pipeline,how", [("", "allreduce"), ("1", "allreduce"), ("1", "rs_ag")])
def test_bench_rank_under_rccl(pipeline, how):
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()),
               HARK_ALLREDUCE=how, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    if pipeline:
        env["HARK_FORCE_PIPELINE"] = "1"
        env["HARK_PRODUCER_WGS"] = "240"                      # what N > 1 runs use: the producer leaves 16 CUs to RCCL
        env["HARK_OVERLAP"] = "1"                             # all three dimensions pinned: nothing is measured at start-up
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--rows", "3000000", "--groups", str(1 << 20),
                          "--steps", "5", "--warmup", "2", "--cpu-rows", "0", "--configs", "0", "--pmc", "0"],
                         capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0, out.stdout + out.stderr
    line = json.loads(out.stdout.strip().splitlines()[-1])
    assert line["rccl_ranks"] == 1 and line["n_gpus"] == 1
    assert line["check"]["count_checksum"] is True and line["check"]["sum_checksum"] is True
    assert line["check"]["tolerance_variant"]["within_bar"] is True
    assert line["config"]["pipelined_steps"] == bool(pipeline)
    assert line["config"]["producer_workgroups"] == (240 if pipeline else "all CUs")
    assert line["config"]["merge"].startswith("RCCL")
    assert len(line["ms_per_step_by_rank"]) == 1 and line["value"] > 0
    # one rank: the strong- and the weak-scaling table coincide -- ONE run, labelled with the metric's own configuration
    assert line["scaling"] == "strong" and line["weak"] is None
    assert line["config"]["rows_total"] == 3000000 and line["config"]["rows_per_gpu"] == 3000000 and line["config"]["mode"] == "single"
    assert line["config"]["measured_at_startup_ms_per_step"] is None


def test_bench_rank_measures_its_producer_geometry_at_startup():
    """With peers (here: one rank, HARK_FORCE_PIPELINE) and nothing pinned, the rank times 240 workgroups pipelined, all CUs
    pipelined and all CUs serial, each with both forms of the merge, before the warm-up and runs the fastest; the line says
    what was measured and chosen."""
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()),
               HARK_FORCE_PIPELINE="1", HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    env.pop("HARK_PRODUCER_WGS", None)
    env.pop("HARK_OVERLAP", None)
    env.pop("HARK_ALLREDUCE", None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--rows", "20000000", "--groups", str(1 << 20),
                          "--steps", "5", "--warmup", "2", "--cpu-rows", "0", "--configs", "0", "--pmc", "0"],
                         capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0, out.stdout + out.stderr
    line = json.loads(out.stdout.strip().splitlines()[-1])
    cfg = line["config"]
    m = cfg["measured_at_startup_ms_per_step"]
    assert set(m) == {g + ", " + h for g in ("240 workgroups, pipelined", "all CUs, pipelined", "all CUs, serial") for h in ("allreduce", "rs_ag")}, m
    assert all(x > 0 for x in m.values()), m
    import bench
    best, margin = bench.pick_candidate(m)                                # the default unless another wins by > 3 %
    assert abs(cfg["startup_choice_margin_over_default"] - margin) < 1e-12
    assert cfg["pipelined_steps"] == (", pipelined" in best) and cfg["overlap"] == (", pipelined" in best)
    assert cfg["producer_workgroups"] == (240 if best.startswith("240") else "all CUs")
    assert cfg["allreduce"] == best.rsplit(", ", 1)[1]
    assert line["check"]["count_checksum"] is True and line["check"]["sum_checksum"] is True


def test_bench_line_compares_hip_with_the_cpu_port_and_measures_traffic():
    """The default single-GPU line at a size where the partition path's byte model holds: the HIP result on the CPU
    sample's rows equals the port's (keys, sums, counts), and -- where rocprofv3 exists -- the HBM traffic comes from
    --pmc child runs of this very bench.py, counts full passes only (steps + warm-up of the child, no setup launch) and
    matches the model of a write-once / read-once partition: 12 B/row read + 6 B per surviving row written and read back
    = 12 + 12 x selectivity B/row (a 0.75x dilution by a tiny setup launch went unnoticed under a 0.9x-3x bar)."""
    import shutil
    have = shutil.which("rocprofv3") is not None
    rows = 128_000_000
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--rows", str(rows), "--groups", str(1 << 20), "--steps", "3", "--warmup", "1",
                          "--cpu-rows", "2000000", "--configs", "0", "--pmc", "1" if have else "0"],
                         capture_output=True, text=True, timeout=900, env=env)
    assert out.returncode == 0, out.stdout + out.stderr
    line = json.loads(out.stdout.strip().splitlines()[-1])
    assert line["check"]["hip_equals_cpu_port_on_sample"] is True, line["check"]
    assert line["check"]["hip_vs_cpu_port"]["rows"] == 2000000
    assert line["cpu_baseline"]["kind"] == "port" and line["cpu_baseline"]["cores"] == 1
    tv = line["check"]["tolerance_variant"]                                  # the uniform [0,1) value column, one untimed pass
    assert tv["within_bar"] is True and tv["rows"] == rows and tv["max_relative_error_vs_f64_scatter_add"] <= 1e-5, tv
    assert tv["hip_vs_cpu_port"]["within_bar"] is True and tv["hip_vs_cpu_port"]["counts_equal"] is True, tv
    if have:
        r = line["roofline"]
        assert r["traffic_measured_in_run"] is True, r.get("traffic_note")
        model = (12.0 + 12.0 * 0.5) * rows
        assert 0.9 * model < r["traffic"] < 1.1 * model, (r["traffic"], model, r["traffic_by_kernel"])
        for kn, dd in r["traffic_by_kernel"].items():
            assert dd["launches_profiled"] == 3, (kn, dd)                          # 2 steps + 1 warm-up of the child
            assert dd["FETCH_SIZE_launches_dropped_as_small"] == 0 and dd["WRITE_SIZE_launches_dropped_as_small"] == 0, (kn, dd)


def test_strong_shard_of_the_billion_row_table_under_rccl():
    """What rank 3 of 8 runs in the strong-scaling mode -- rows shard_range(1e9, 3, 8) of the 1B-row table, generated from
    their GLOBAL row numbers -- as one RCCL rank: the plan geometry of a 1.25e8-row shard, the merge, the checksums."""
    import bench
    from harkdb_amd.dist import shard_range
    (label, n_local, first_row, rows_total), _ = bench.mode_specs(10**9, 3, 8)
    assert (label, n_local, first_row, rows_total) == ("strong", 125_000_000, 375_000_000, 10**9) == ("strong",) + (shard_range(10**9, 3, 8)[1] - shard_range(10**9, 3, 8)[0], shard_range(10**9, 3, 8)[0], 10**9)
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()),
               HARK_FORCE_PIPELINE="1", HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--rows", str(n_local), "--groups", str(1 << 20),
                          "--first-row", str(first_row),
                          "--steps", "10", "--warmup", "3", "--cpu-rows", "0", "--configs", "0", "--pmc", "0", "--tolerance-check", "0"],
                         capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0, out.stdout + out.stderr
    line = json.loads(out.stdout.strip().splitlines()[-1])
    assert line["check"] == {"count_checksum": True, "sum_checksum": True}
    assert line["config"]["rows_this_rank"] == n_local and line["config"]["first_row_of_rank0"] == first_row and line["roofline"]["frac"] > 0.2, line["roofline"]


def test_a_dying_configs_child_does_not_lose_the_headline():
    """The extra configs run in a child process with a wall-clock budget; a child that hangs mid-run (here: told to stall after
    its third config, as a wedged kernel would) is killed at the budget and costs its unfinished configs only -- the ONE line
    still carries the headline and the configs the child had written."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    env["HARK_BENCH_CHILD_STALL"] = "G4096"
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--rows", "16000000", "--groups", str(1 << 20), "--steps", "2", "--warmup", "1",
                          "--cpu-rows", "0", "--configs", "1", "--config-scale", "0.02", "--configs-budget", "30", "--pmc", "0"],
                         capture_output=True, text=True, timeout=900, env=env)
    assert out.returncode == 0, out.stdout + out.stderr
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout
    line = json.loads(lines[0])
    assert line["value"] > 0 and line["check"]["count_checksum"] is True and line["roofline"]["frac"] > 0
    assert "killed at its wall-clock budget" in line["configs"]["error"], line["configs"]
    assert "C3_no_filter" in line["configs"] and "G16" in line["configs"] and "G4096" in line["configs"]     # what it had finished survives
    assert "G13000" not in line["configs"] and "C5_three_aggregates_all_groups" not in line["configs"]


def test_a_crashing_configs_child_does_not_lose_the_headline():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--rows", "16000000", "--groups", str(1 << 20), "--steps", "2", "--warmup", "1",
                          "--cpu-rows", "0", "--configs", "1", "--config-scale", "-1", "--pmc", "0"],          # a negative scale: the child raises
                         capture_output=True, text=True, timeout=900, env=env)
    assert out.returncode == 0, out.stdout + out.stderr
    line = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][0])
    assert line["value"] > 0 and "exited with rc" in line["configs"]["error"], line["configs"]


def test_bench_line_survives_sigterm_during_the_extras():
    """A driver that gives up while the extras run (SIGTERM) still gets the line: the headline is printed by the handler."""
    import signal
    import time
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    pr = subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py"), "--rows", "64000000", "--groups", str(1 << 20), "--steps", "2", "--warmup", "1",
                           "--cpu-rows", "150000000", "--configs", "0", "--pmc", "0", "--tolerance-check", "0"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env)
    # the CPU port over 6.4e7 rows takes seconds: wait until the GPU part is over (the oracle library gets loaded), then give up
    deadline = time.time() + 300
    while time.time() < deadline and pr.poll() is None:
        try:
            maps = open(f"/proc/{pr.pid}/maps").read()
        except OSError:
            break
        if "liboracle" in maps:
            break
        time.sleep(0.2)
    if pr.poll() is None:
        pr.send_signal(signal.SIGTERM)
    so, se = pr.communicate(timeout=300)
    lines = [ln for ln in so.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, so + se
    line = json.loads(lines[0])
    assert line["value"] > 0 and line["check"]["count_checksum"] is True


def test_bench_extra_configs_at_a_small_scale():
    """Every `configs.*` workload of the bench line at 2 % of its size: none errors, the sparse-key GROUP BY takes the hash
    path and equals the dense result after key mapping, the join finds the expected pairs."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--rows", "6000000", "--groups", str(1 << 20), "--steps", "2", "--warmup", "1",
                          "--cpu-rows", "0", "--configs", "1", "--config-scale", "0.02", "--pmc", "0"],
                         capture_output=True, text=True, timeout=900, env=env)
    assert out.returncode == 0, out.stdout + out.stderr
    line = json.loads(out.stdout.strip().splitlines()[-1])
    cfg = line["configs"]
    assert "error" not in cfg, cfg
    # the LAST key of the line is the tail-safe summary ({config: [ms, fraction of peak]}): whoever keeps only the end of stdout
    # still sees the configs that sit at the front of `configs`
    assert list(line)[-1] == "summary" and out.stdout.rstrip().endswith("}}")
    tail = out.stdout.rstrip()[-4096:]
    for name in ("headline", "C3_no_filter", "G16", "G4096", "G13000", "SWEEP_selectivity_x_groups.G2^20_sel1.0", "REF_join_u32", "C4_join_share", "C0_reference_csv.query_groupby"):
        assert '"' + name + '"' in tail, name
        assert line["summary"][name][0] > 0
    assert len(json.dumps(line["summary"])) <= 2048
    assert cfg["C3_no_filter"]["count_checksum"] is True and cfg["C3_no_filter"]["sum_checksum"] is True
    for name in ("HEADLINE_sorted_keys", "C3_no_filter_sorted_keys"):                   # a table kept in key order: the window path, same survivors
        assert cfg[name]["count_checksum"] is True and cfg[name]["path"].startswith("window"), cfg[name]
    assert "G2^20_sel1.0" in cfg["SWEEP_selectivity_x_groups"] and cfg["SWEEP_selectivity_x_groups"]["G2^20_sel1.0"]["count_checksum"] is True
    for name in ("C3_no_filter", "G16", "G4096", "G13000", "SWEEP_selectivity_x_groups", "SPARSE_groupby", "SPARSE_five_aggregates", "C2_filter_proj", "C1_projection", "REF_query_groupby_dense",
                 "REF_query_groupby_hash", "ORDER_BY", "ORDER_BY_32bit", "ORDER_BY_i64", "ORDER_BY_sorted_column", "REF_join_u32", "REF_join_u32_sorted_probe", "C4_join_share", "C5_pipeline_share",
                 "C5_three_aggregates", "C5_three_aggregates_all_groups"):
        assert name in cfg, name
    sp = cfg["SPARSE_groupby"]
    assert sp["path"] == "hash" and sp["equals_dense_result_after_key_mapping"] is True and sp["result_shape"][0] > 900_000, sp
    assert cfg["REF_query_groupby_dense"]["path"] == "dense" and cfg["REF_query_groupby_hash"]["path"] == "hash"
    assert cfg["C4_join_share"]["pairs"] == cfg["C4_join_share"]["pairs_expected"]
    assert cfg["REF_join_u32"]["pairs"] == cfg["REF_join_u32"]["pairs_expected"] > 0
    assert cfg["REF_join_u32_sorted_probe"]["pairs"] == cfg["REF_join_u32"]["pairs_expected"] and cfg["REF_join_u32_sorted_probe"]["path"].startswith("clustered")   # the probe table in key order: the search path, the same pairs


def test_two_bench_ranks_on_one_gpu_measure_strong_and_weak():
    """The N = 2 invocation with REAL kernels on both ranks (they share the box's one GPU; gloo carries the collectives, staged
    through the host, because RCCL refuses two ranks on one device): the strong-scaling run first -- each rank a row range of
    the SAME seeded table, counts and sums of the merged result equal to the table's -- then the weak run beside it, each with
    its own start-up measurement.  Everything of the 8-GPU run except RCCL between devices."""
    port = _free_port()
    rows = 40_000_000
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK="0", WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HARK_DIST_BACKEND="gloo")
        for k in ("HARK_PRODUCER_WGS", "HARK_OVERLAP", "HARK_ALLREDUCE", "HARK_FORCE_PIPELINE"):
            env.pop(k, None)
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--rows", str(rows), "--groups", str(1 << 20),
                                       "--steps", "3", "--warmup", "1"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env))
    outs = [p.communicate(timeout=900) for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(o[0] + o[1] for o in outs)
    lines = [ln for ln in outs[0][0].splitlines() if ln.startswith("{")]
    assert len(lines) == 1 and not [ln for ln in outs[1][0].splitlines() if ln.startswith("{")], outs
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["rccl_ranks"] == 2 and d["scaling"] == "strong"
    c = d["config"]
    assert c["rows_total"] == rows and c["rows_per_gpu"] == rows // 2 and c["rows_this_rank"] == rows // 2 and c["mode"] == "strong"
    assert d["check"]["count_checksum"] is True and d["check"]["sum_checksum"] is True and d["check"]["all_rows_seen"] is True
    assert set(c["measured_at_startup_ms_per_step"]) == {g + ", " + h for g in ("240 workgroups, pipelined", "all CUs, pipelined", "all CUs, serial") for h in ("allreduce", "rs_ag")}
    w = d["weak"]
    assert w["scaling"] == "weak" and w["config"]["rows_total"] == 2 * rows and w["config"]["rows_per_gpu"] == rows
    assert w["check"]["count_checksum"] is True and w["check"]["sum_checksum"] is True and w["check"]["all_rows_seen"] is True
    assert w["roofline"]["frac"] > 0 and d["roofline"]["frac"] > 0 and len(d["ms_per_step_by_rank"]) == 2
    assert abs(d["value"] - rows / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"] and abs(w["value"] - 2 * rows / (w["ms_per_step"] * 1e-3)) < 1e-6 * w["value"]
    assert "cpu_baseline" in d and d["cpu_baseline"] is None and "configs" not in d            # N > 1: no extras

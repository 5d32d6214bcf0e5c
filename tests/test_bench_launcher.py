"""bench.py's rank protocol without a GPU: `python bench.py --gpus N` bare starts its own N ranks (the driver may run it
that way), and under torchrun it reads RANK / WORLD_SIZE; both give ONE JSON line from rank 0.  --stub 1 swaps the HIP
step for a numpy aggregate and RCCL for gloo -- everything else (launcher, barrier, MAX-over-ranks, JSON) is the code
the GPU run uses."""
import json
import os
import socket
import subprocess
import sys

import pytest

from conftest import ROOT


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _env():
    env = dict(os.environ, HARK_DIST_BACKEND="gloo")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR"):
        env.pop(k, None)
    return env


def _one_json_line(stdout):
    lines = [ln for ln in stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, stdout
    return json.loads(lines[0])


@pytest.mark.parametrize("n", [1, 2, 8])
def test_bare_invocation_launches_its_own_ranks(n):
    """BASELINE.json quotes ONE 1B-row table at 1/2/4/8 GPUs: with N > 1 the line's value / ms_per_step are the STRONG-scaling
    run (rows_total = --rows whatever N is, every rank a row range of the same seeded table) and the weak-scaling run of the
    same invocation (--rows per rank) is the `weak` sub-record."""
    rows = 20000
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--stub", "1", "--rows", str(rows), "--groups", "64",
                          "--steps", "2", "--warmup", "1"], capture_output=True, text=True, timeout=300, env=_env())
    assert out.returncode == 0, out.stdout + out.stderr
    d = _one_json_line(out.stdout)
    assert d["n_gpus"] == n and d["rccl_ranks"] == n and d["steps"] == 2 and d["warmup"] == 1
    assert len(d["ms_per_step_by_rank"]) == n and d["check"]["count_checksum"] is True
    assert d["scaling"] == "strong" and d["higher_is_better"] is True
    assert d["config"]["rows_total"] == rows and d["config"]["rows_per_gpu"] == rows // n
    assert d["check"]["rows_seen_by_all_ranks"] == rows
    assert d["check"]["survivors"] == 9906                     # the SAME table at every N: a function of the global row index only
    assert abs(d["value"] - rows / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]
    if n == 1:
        assert d["weak"] is None                               # strong and weak coincide: one run
    else:
        w = d["weak"]
        assert w["scaling"] == "weak" and w["config"]["rows_per_gpu"] == rows and w["config"]["rows_total"] == rows * n
        assert w["check"]["count_checksum"] is True and w["check"]["rows_seen_by_all_ranks"] == rows * n
        assert len(w["ms_per_step_by_rank"]) == n
        assert abs(w["value"] - rows * n / (w["ms_per_step"] * 1e-3)) < 1e-6 * w["value"]


def test_modes_can_be_selected():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--stub", "1", "--rows", "20000", "--groups", "64",
                          "--steps", "2", "--warmup", "1", "--modes", "weak"], capture_output=True, text=True, timeout=300, env=_env())
    assert out.returncode == 0, out.stdout + out.stderr
    d = _one_json_line(out.stdout)
    assert d["scaling"] == "weak" and d["config"]["rows_total"] == 40000 and d["weak"] is None


def test_strong_shards_tile_the_table():
    import bench
    for rows, world in ((10**9, 8), (10**9, 3), (1000003, 4), (7, 8)):
        specs = [bench.mode_specs(rows, r, world) for r in range(world)]
        strong = [s[0] for s in specs]
        assert all(s[0] == "strong" and s[3] == rows for s in strong)
        assert strong[0][2] == 0 and sum(s[1] for s in strong) == rows
        assert all(strong[r][2] + strong[r][1] == strong[r + 1][2] for r in range(world - 1))       # contiguous, in rank order
        assert all(s[2] % 4 == 0 for s in strong if s[1])                                                     # 16-byte loads stay aligned
        weak = [s[1] for s in specs]
        assert all(w == ("weak", rows, r * rows, rows * world) for r, w in enumerate(weak))
    assert bench.mode_specs(123, 0, 1) == [("single", 123, 0, 123)]


def test_startup_choice_keeps_the_default_on_near_ties():
    """six timed steps cannot tell near-ties apart; the default candidate stays unless another wins by > 3 %, and
    pinning one dimension (HARK_PRODUCER_WGS / HARK_OVERLAP / HARK_ALLREDUCE) leaves the others measured."""
    import bench
    assert bench.pick_candidate({"a": 1.00, "b": 0.98, "c": 1.2}) == ("a", 0.0)
    name, margin = bench.pick_candidate({"a": 1.00, "b": 0.90, "c": 0.95})
    assert name == "b" and abs(margin - 0.10) < 1e-12
    full = bench.tuning_candidates({})
    assert full[0] == (240, True, "allreduce") and len(full) == 6
    assert bench.tuning_candidates({"HARK_PRODUCER_WGS": "0"}) == [(0, True, "allreduce"), (0, False, "allreduce"), (0, True, "rs_ag"), (0, False, "rs_ag")]
    assert bench.tuning_candidates({"HARK_OVERLAP": "0"}) == [(0, False, "allreduce"), (0, False, "rs_ag")]
    assert bench.tuning_candidates({"HARK_ALLREDUCE": "rs_ag"}) == [(240, True, "rs_ag"), (0, True, "rs_ag"), (0, False, "rs_ag")]
    assert bench.tuning_candidates({"HARK_PRODUCER_WGS": "240", "HARK_OVERLAP": "1", "HARK_ALLREDUCE": "allreduce"}) == [(240, True, "allreduce")]


def test_under_torchrun():
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                          "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--stub", "1", "--rows", "20000",
                          "--groups", "64", "--steps", "2", "--warmup", "1"], capture_output=True, text=True, timeout=300, env=_env())
    assert out.returncode == 0, out.stdout + out.stderr
    d = _one_json_line(out.stdout)
    assert d["n_gpus"] == 2 and d["rccl_ranks"] == 2 and d["backend"] == "gloo"
    assert d["scaling"] == "strong" and d["config"]["rows_total"] == 20000 and d["weak"]["config"]["rows_total"] == 40000


def test_world_size_mismatch_fails_loudly():
    env = dict(_env(), RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--stub", "1", "--rows", "1000", "--groups", "8"],
                         capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode != 0 and "WORLD_SIZE=1" in (out.stdout + out.stderr)


def test_traffic_mean_counts_full_passes_only():
    """measure_traffic's launch filter (a 65536-row setup launch diluted the per-launch mean to 0.75x) and
    its refusal to nest a profiler."""
    import bench
    full = [5859458.0, 5859460.0, 5859455.0]
    assert bench.headline_launches(full + [466.0]) == full
    assert bench.headline_launches(full) == full
    assert bench.headline_launches([466.0]) == [466.0]
    assert bench.under_profiler({"ROCP_TOOL_LIBRARIES": "x"}) and bench.under_profiler({"LD_PRELOAD": "/opt/rocm/lib/librocprofiler-sdk-tool.so"})
    assert not bench.under_profiler({"PATH": "/usr/bin"})

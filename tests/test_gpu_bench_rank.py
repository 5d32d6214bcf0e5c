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
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--rows", "3000000", "--groups", str(1 << 20),
                          "--steps", "5", "--warmup", "2", "--cpu-rows", "0", "--configs", "0", "--pmc", "0"],
                         capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0, out.stdout + out.stderr
    line = json.loads(out.stdout.strip().splitlines()[-1])
    assert line["rccl_ranks"] == 1 and line["n_gpus"] == 1
    assert line["check"] == {"count_checksum": True, "sum_checksum": True}
    assert line["config"]["pipelined_steps"] == bool(pipeline)
    assert line["config"]["producer_workgroups"] == (240 if pipeline else "all CUs")
    assert line["config"]["merge"].startswith("RCCL")
    assert len(line["ms_per_step_by_rank"]) == 1 and line["value"] > 0


def test_bench_line_compares_hip_with_the_cpu_port_and_measures_traffic():
    """The default single-GPU line at a small size: the HIP result on the CPU sample's rows equals the port's (keys, sums,
    counts), and -- where rocprofv3 exists -- the HBM traffic comes from --pmc child runs of this very bench.py."""
    import shutil
    have = shutil.which("rocprofv3") is not None
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--rows", "8000000", "--groups", str(1 << 20), "--steps", "3", "--warmup", "1",
                          "--cpu-rows", "2000000", "--configs", "0", "--pmc", "1" if have else "0"],
                         capture_output=True, text=True, timeout=900, env=env)
    assert out.returncode == 0, out.stdout + out.stderr
    line = json.loads(out.stdout.strip().splitlines()[-1])
    assert line["check"]["hip_equals_cpu_port_on_sample"] is True, line["check"]
    assert line["check"]["hip_vs_cpu_port"]["rows"] == 2000000
    assert line["cpu_baseline"]["kind"] == "port" and line["cpu_baseline"]["cores"] == 1
    if have:
        r = line["roofline"]
        assert r["traffic_measured_in_run"] is True, r.get("traffic_note")
        assert 12.0 * 8000000 * 0.9 < r["traffic"] < 12.0 * 8000000 * 3.0          # at least the three columns, at most 3x

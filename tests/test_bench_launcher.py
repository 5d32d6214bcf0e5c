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
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--stub", "1", "--rows", "20000", "--groups", "64",
                          "--steps", "2", "--warmup", "1"], capture_output=True, text=True, timeout=300, env=_env())
    assert out.returncode == 0, out.stdout + out.stderr
    d = _one_json_line(out.stdout)
    assert d["n_gpus"] == n and d["rccl_ranks"] == n and d["steps"] == 2 and d["warmup"] == 1
    assert len(d["ms_per_step_by_rank"]) == n and d["check"]["count_checksum"] is True
    assert d["scaling"] == "weak" and d["higher_is_better"] is True


def test_under_torchrun():
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                          "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--stub", "1", "--rows", "20000",
                          "--groups", "64", "--steps", "2", "--warmup", "1"], capture_output=True, text=True, timeout=300, env=_env())
    assert out.returncode == 0, out.stdout + out.stderr
    d = _one_json_line(out.stdout)
    assert d["n_gpus"] == 2 and d["rccl_ranks"] == 2 and d["backend"] == "gloo"


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

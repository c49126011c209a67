"""CPU-side checks of bench.py's measurement plumbing (round 5): the telemetry windows, the median, the
self-spawn command line and the process-group report over gloo - none of it needs a GPU."""
import json
import os
import subprocess
import sys
import time

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402  (no torch import, no GPU touched at import time)


def test_median_and_telemetry_windows():
    assert bench.median([3.0, 1.0, 2.0]) == 2.0 and bench.median([4.0, 1.0, 2.0, 3.0]) == 2.5
    tel = bench.Telemetry.__new__(bench.Telemetry)  # no device lookup: feed samples by hand
    tel.samples, tel.period, tel.source, tel._files, tel._thread = [], 0.02, "synthetic", None, None
    t0 = time.perf_counter()
    for i in range(10):
        tel.samples.append((t0 + 0.1 * i, 2000.0 + i, 1000.0 + 10 * i, 50.0))
    tel.samples.append((t0 + 0.35, None, None, None))  # a sample whose files could not be read
    w = tel.window(t0 + 0.25, t0 + 0.65)  # samples 3, 4, 5, 6 and the empty one
    assert w["samples"] == 5 and w["sclk_mhz_min"] == 2003.0 and w["sclk_mhz_max"] == 2006.0
    assert w["sclk_mhz_median"] in (2004.0, 2005.0) and w["power_w_median"] in (1040.0, 1050.0)
    empty = tel.window(t0 + 5, t0 + 6)
    assert empty["samples"] == 0 and empty["sclk_mhz_median"] is None and empty["power_w_median"] is None
    assert tel.describe()["source"] == "synthetic"


def test_telemetry_without_a_device_is_inert():
    """No amdgpu hwmon for device 0 in this container and (normally) no rocm-smi: start / stop / window must be
    harmless no-ops that leave `None` in the line, never an exception."""
    tel = bench.Telemetry(0)
    tel.start()
    time.sleep(0.05)
    tel.stop()
    w = tel.window(0, time.perf_counter() + 1)
    assert set(w) >= {"samples", "sclk_mhz_median", "power_w_median"}
    json.dumps(dict(tel.describe(), sustained=w))  # serialisable


def test_self_spawn_launches_torchrun_with_the_same_arguments(monkeypatch):
    seen = {}

    def fake_call(cmd, env=None):
        seen["cmd"], seen["env"] = cmd, env
        return 0
    monkeypatch.setattr(subprocess, "call", fake_call)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "20", "--warmup", "5"])
    bench.self_spawn(4)
    cmd = seen["cmd"]
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nnodes=1" in cmd
    assert cmd[cmd.index("--nproc-per-node") + 1] == "4" and cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert 1024 < int(cmd[cmd.index("--master-port") + 1]) < 65536
    assert cmd[-6:] == ["--gpus", "4", "--steps", "20", "--warmup", "5"] and cmd[-7].endswith("bench.py")
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    monkeypatch.setattr(subprocess, "call", lambda cmd, env=None: 3)
    with pytest.raises(SystemExit) as e:
        bench.self_spawn(2)
    assert e.value.code == 3


_REPORT_WORKER = r"""
import json, os, sys
sys.path.insert(0, %(root)r)
import torch, torch.distributed as dist
import bench
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()

class Lib:  # what dist_report reads from the library
    def sp_window_bits(self): return 21
    def sp_table_bytes(self): return 3 << 30

info = bench.dist_report(torch, dist, torch.device("cpu"), 0, world, rank, False, 1000.0 + rank, Lib())
if rank == 0:
    print("REPORT " + json.dumps(info))
dist.barrier()
dist.destroy_process_group()
"""


def test_process_group_report_over_gloo(tmp_path):
    """bench.dist_report with world size 2 on CPU: collective on every rank, rank 0 gets both ranks' entries; the
    device queries fail without a GPU and must degrade to an `error` entry, not an exception."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    script = tmp_path / "worker.py"
    script.write_text(_REPORT_WORKER % {"root": ROOT})
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", str(port), str(script)],
                         capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("REPORT ")][0]
    info = json.loads(line[len("REPORT "):])
    assert info["backend"] == "gloo" and info["world_size"] == 2 and len(info["ranks"]) == 2
    assert [r["rank"] for r in info["ranks"]] == [0, 1]
    assert info["rccl_version"] is None  # gloo
    for r in info["ranks"]:  # no GPU here: either a full entry or the degraded one
        assert "error" in r or r["window_bits"] == 21

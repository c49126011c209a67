"""CPU-side checks of bench.py's measurement plumbing (benchlib/): the ONE line's size and completeness, the telemetry
windows, the median, the self-spawn command line, the roofline file lists and the process-group report over gloo - none of
it needs a GPU."""
import json
import os
import subprocess
import sys
import time

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402  (no torch import, no GPU touched at import time)
from benchlib import common, launch, line, roofline, telemetry  # noqa: E402


def _canned_detail():
    """A full-size result of the default run: round 5's 20 KB line (the one the driver could not parse), which is
    exactly the `detail` dict bench.py now hands to benchlib.line.main_line."""
    return json.load(open(os.path.join(ROOT, "profiles", "r05_bench_default.json")))


def test_line_is_small_and_complete():
    """the printed line is one strictly valid JSON object of less than 8 KB with every key of the
    contract, a numbers-only roofline, cpu_baseline, a small airfri and `summary` last."""
    detail = _canned_detail()
    assert len(json.dumps(detail)) > 16000  # the canned input really is the oversized one
    out = line.main_line(detail)
    text = line.check_line(out)
    assert len(text.encode()) < 8192 and "\n" not in text
    back = json.loads(text, parse_constant=lambda c: pytest.fail("non-finite constant %s in the line" % c))
    for k in line.REQUIRED_KEYS:
        assert k in back, k
    assert back["metric"] == "pedersen_hashes_per_sec" and back["unit"] == "hashes/s" and back["dtype"] == "u32x9"
    assert back["value"] == pytest.approx(detail["value"], rel=1e-8)
    assert back["ms_per_step"] == pytest.approx(detail["ms_per_step"], rel=1e-8)
    assert back["value"] == pytest.approx(back["config"]["hashes_per_step"] * back["steps"] / back["timed"]["median_s"], rel=1e-6)
    assert back["config"]["workload"].startswith("2^16-leaf") and "model" not in back["config"]
    r = back["roofline"]
    for k in line.ROOFLINE_KEYS + ("kernel", "instr_per_hash", "avg_launch_us", "frac_at_held_clock", "frac_at_2_cycle_peak",
                                   "algorithmic_bytes_per_launch", "hbm", "whole_region", "sources"):
        assert k in r, k
    assert r["frac"] == pytest.approx(r["achieved"] / r["peak"], rel=1e-6)
    assert r["hbm"]["bound"] == "hbm" and r["hbm"]["frac"] == pytest.approx(r["hbm"]["achieved"] / r["hbm"]["peak"], rel=1e-6)
    assert r["algorithmic_bytes_per_launch"] == 96 * int(r["hashes_per_launch"])
    assert all(isinstance(v, (int, float, bool, type(None))) or k in ("bound", "kernel", "unit", "traffic_unit", "hbm",
                                                                      "whole_region", "sources")
               for k, v in r.items()), "prose crept back into the roofline"
    assert all(src.startswith("profiles/") for src in r["sources"])
    cb = back["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 0 and len(cb["sample"]) <= 200
    assert len(json.dumps(back["airfri"])) <= 1200 and back["airfri"]["commits_per_sec"] > 0
    assert back["airfri"]["cpu_baseline"]["commits_per_sec_scaled_to_2p20"] > 0
    assert list(back)[-1] == "summary"
    assert back["summary"]["pedersen_hashes_per_sec"] == back["value"]
    assert back["summary"]["airfri_commits_per_sec"] == back["airfri"]["commits_per_sec"]
    assert back["detail"] == line.DETAIL_FILE


def test_line_with_a_process_group_keeps_the_rccl_proof(tmp_path):
    """at N > 1 the main line still says what RCCL saw (world size, version, per-rank value, the
    combine check) - and stays under the bound with an 8-rank report attached."""
    detail = _canned_detail()
    detail["n_gpus"] = 8
    detail["combine_matches_recomputed"] = True
    detail["dist"] = {"backend": "nccl", "world_size": 8, "rccl_version": "2.26.6", "forced_at_one_gpu": False,
                      "visible_devices": 8, "per_rank_value": {"min": 7.7e8, "max": 7.9e8, "unit": "hashes/s ..."},
                      "ranks": [{"rank": i, "device_index": i, "name": "AMD Instinct MI355X", "pci": "0000:%02x:00.0" % (16 * i + 5),
                                 "free_hbm_gib": 190.0, "total_hbm_gib": 287.9, "window_bits": 26, "table_gib": 75.0,
                                 "pid": 1000 + i, "cpus_allowed": 128, "local_hashes_per_sec": 7.8e8} for i in range(8)],
                      "peer_access": [[1] * 8 for _ in range(8)], "env": {"HSA_ENABLE_IPC_MODE_LEGACY": "0"},
                      "link_types": {"system": {"(Topology) Link type between DRM devices %d and %d" % (a, b): "XGMI"
                                                for a in range(8) for b in range(a + 1, 8)}}}
    out = line.main_line(detail)
    text = line.check_line(out)
    assert len(text.encode()) < 8192
    d = json.loads(text)["dist"]
    assert d["world_size"] == 8 and d["rccl_version"] == "2.26.6" and d["backend"] == "nccl"
    assert d["per_rank_value"] == {"min": 7.7e8, "max": 7.9e8} and d["ranks_reported"] == 8 and len(d["devices"]) == 8
    assert d["peer_access_all"] is True and json.loads(text)["combine_matches_recomputed"] is True
    # emit(): the line on stdout, the detail in its file
    import io
    buf = io.StringIO()
    os.environ["STARKPERP_BENCH_DETAIL"] = str(tmp_path / "detail.json")
    try:
        printed = line.emit(out, detail=detail, stream=buf)
    finally:
        del os.environ["STARKPERP_BENCH_DETAIL"]
    assert buf.getvalue() == printed + "\n" and buf.getvalue().count("\n") == 1
    assert json.load(open(tmp_path / "detail.json"))["dist"]["ranks"][7]["rank"] == 7


def test_emit_never_prints_an_oversized_line(capsys):
    """A line that breaks the bound (somebody adds prose again) is cut to the contract's keys rather than printed."""
    import io
    out = line.main_line(_canned_detail())
    out["config"]["essay"] = "x" * 9000
    with pytest.raises(ValueError):
        line.check_line(out)
    buf = io.StringIO()
    printed = line.emit(out, stream=buf)
    assert len(printed.encode()) < 8192
    back = json.loads(printed)
    assert back["value"] > 0 and back["roofline"]["frac"] > 0 and back["cpu_baseline"]["value"] > 0
    assert "limit 8192" in capsys.readouterr().err


def test_non_finite_numbers_become_null():
    out = line._clean({"a": float("nan"), "b": [float("inf"), 1.0], "c": {"d": -float("inf")}, "e": 0.1234567891234})
    assert out == {"a": None, "b": [None, 1.0], "c": {"d": None}, "e": 0.123456789}


def test_roofline_files_put_this_round_first_and_stamp_nothing_stale():
    """the r06 passes come first in every list, and an older file is never called "the same
    configuration as this run" unless its recorded config_key equals this run's."""
    for files in (roofline.PMC_FILES, roofline.AIRFRI_PMC_FILES, roofline.VALU_ISSUE_FILES):
        assert files[0].startswith("r06_") and list(files) == sorted(files, reverse=True)
    key = roofline.merkle_config_key(20, [20], 2, 26)
    assert key == "merkle:steps=20:calls=20:streams=2:w=26"
    t = roofline.pmc_traffic("sp::ped_accumulate_kernel", key)
    assert t["source"].startswith("profiles/r0") and t["same_configuration_as_this_run"] is (t["config_key"] == key)
    assert roofline.pmc_traffic("sp::ped_accumulate_kernel", "merkle:steps=7:calls=7:streams=1:w=21")[
        "same_configuration_as_this_run"] is False
    a = roofline.pmc_traffic("sp::ped_accumulate_kernel", roofline.airfri_config_key(26), roofline.AIRFRI_PMC_FILES)
    assert a is None or a["same_configuration_as_this_run"] is (a["config_key"] == roofline.airfri_config_key(26))
    r = roofline.valu_issue(1.0e9, 26, "test", include_finish=False)
    assert r["instr_source"].startswith("profiles/r0") and r["frac"] == pytest.approx(r["achieved"] / r["peak"])
    held = roofline.add_held_clock(dict(r), 2000.0)
    assert held["frac_at_held_clock"] == pytest.approx(r["frac"] * 2400.0 / 2000.0)


def test_bench_py_stays_small_and_delegates():
    """bench.py keeps argument parsing, the timed region, the CPU legs and the print."""
    src = open(os.path.join(ROOT, "bench.py")).read().splitlines()
    assert len(src) < 600
    for name in ("Telemetry", "self_spawn", "dist_report", "median"):  # re-exported for tools that import bench
        assert hasattr(bench, name)
    assert bench.median is common.median and bench.Telemetry is telemetry.Telemetry and bench.self_spawn is launch.self_spawn
    # only bench.py itself (its cpu_baseline legs) may import the oracle: nothing under benchlib/ does
    import glob
    for f in glob.glob(os.path.join(ROOT, "benchlib", "*.py")):
        text = open(f).read()
        assert "from oracle" not in text and "import oracle" not in text, f


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
from benchlib import launch as bench
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


def test_plan_splits_steps_into_the_fewest_even_calls():
    from benchlib import merkle
    assert merkle.plan(20, 64) == [20] and merkle.plan(128, 64) == [64, 64] and merkle.plan(130, 64) == [44, 43, 43]
    assert merkle.plan(0, 64) == [] and merkle.plan(5, 1) == [1] * 5
    for k, cap in ((20, 64), (77, 16), (1000, 64)):
        p = merkle.plan(k, cap)
        assert sum(p) == k and max(p) <= cap and max(p) - min(p) <= 1 and len(p) == -(-k // cap)


def test_kernel_roofline_arithmetic_from_a_canned_measurement():
    """The roofline object of the headline kernel from (total ms, launches, hashes) of sp_profile_end: achieved =
    instr_per_hash x hashes / 64 / time, frac against 1024 SIMDs x 2.4 GHz / c_mix, the held-clock variant, HBM from the
    96 algorithmic bytes per hash, whole_region priced at the whole forest's measured instruction count."""
    from benchlib import merkle
    launches, hashes_per_launch, avg_us = 3562, 491520, 423.5
    prof = (launches * avg_us / 1e3, launches, launches * hashes_per_launch)
    r = merkle.kernel_roofline(prof, 26, roofline.merkle_config_key(20, [20], 2, 26), 7.9e8, 2330.0, 0.75)
    per_hash = roofline.valu_counts(26)[0]
    rate = hashes_per_launch / (avg_us * 1e-6)
    assert r["bound"] == "valu_issue" and r["instr_per_hash"] == per_hash
    assert r["achieved"] == pytest.approx(rate * per_hash / 64.0) and r["avg_launch_us"] == pytest.approx(avg_us)
    assert r["peak"] == pytest.approx(1024 * 2.4e9 / roofline.valu_cycles_per_instr())
    assert r["frac"] == pytest.approx(r["achieved"] / r["peak"]) and 0.7 < r["frac"] < 0.9
    assert r["frac_at_held_clock"] == pytest.approx(r["frac"] * 2400.0 / 2330.0)
    assert r["hbm"]["achieved"] == pytest.approx(96 * rate / 1e9) and r["hbm"]["frac"] == pytest.approx(96 * rate / 1e9 / 8000.0)
    w = r["whole_region"]
    f = roofline.forest_instr_per_hash(26)
    assert w["instr_per_hash"] == (f[0] if f else per_hash + roofline.valu_counts(26)[1])
    assert w["achieved"] == pytest.approx(7.9e8 * w["instr_per_hash"] / 64.0) and "held_clock_mhz" not in w
    slim = line.slim_roofline(r, 96 * hashes_per_launch)
    assert slim["kernel"] == "ped_accumulate_kernel" and slim["algorithmic_bytes_per_launch"] == 96 * hashes_per_launch
    assert slim["whole_region"]["frac"] == w["frac"]


def test_c3_roofline_uses_live_rates_where_given():
    c = roofline.c3_roofline({"verify_keyed": 2.4e8, "verify_ladder": 4.5e7, "verify_keyed_2p18": 3.3e8})
    if c is None:
        pytest.skip("no C3 counter file committed yet")
    k = c["verify_keyed"]
    assert k["rate_is"].startswith("live") and k["achieved"] == pytest.approx(k["instr_per_item"] * 2.4e8 / 64.0)
    assert k["frac"] == pytest.approx(k["achieved"] / c["peak"]) and k["waves_per_simd"] == 1.0
    assert c["verify_keyed_2p18"]["frac"] == pytest.approx(k["frac"] * 3.3e8 / 2.4e8)
    assert "verify_ladder_2p18" not in c  # no live rate given: not invented
    for key in ("message_hash_chains", "tree_paths", "verify_keyed_4096"):
        assert c[key]["rate_is"].startswith("the counter pass") and 0 < c[key]["frac"] < 1
        assert 1.5 < c[key]["ns_per_dependent_instr"] < 6

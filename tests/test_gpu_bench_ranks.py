"""bench.py's N > 1 path end to end on a one-GPU box: two ranks launched exactly as the driver does
(torch.distributed.run), both on device 0, sub-roots exchanged over gloo (STARKPERP_BENCH_SHARE_GPU=1,
a hook that exists only for this test).  Checks the JSON contract and the combined root."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _line_and_detail(out, detail_path):
    """(the ONE stdout line parsed, the detail file parsed or None); the line must respect the driver's bound."""
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]  # rank 0 prints ONE line
    assert len(lines[0].encode()) < 8192, len(lines[0])
    d = json.loads(lines[0], parse_constant=lambda c: pytest.fail("non-finite constant %s" % c))
    detail = json.load(open(detail_path)) if os.path.exists(detail_path) else None
    return d, detail


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("workload,extra", [("merkle", ["--steps", "6", "--warmup", "2"]),
                                            ("airfri", ["--steps", "1", "--warmup", "1", "--log-rows", "14"]),
                                            ("airfri", ["--steps", "1", "--warmup", "0", "--log-rows", "19"])])
def test_two_ranks_share_one_gpu(workload, extra, tmp_path):
    env = dict(os.environ, STARKPERP_BENCH_SHARE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0",
               STARKPERP_BENCH_DETAIL=str(tmp_path / "detail.json"), STARKPERP_WINDOW_BITS="16")  # two ranks share one GPU here: small tables whatever the caller set
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"),
           "--gpus", "2", "--workload", workload, "--window-bits", "0", "--no-extras", "--no-cpu-baseline"] + extra
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    d, detail = _line_and_detail(out, str(tmp_path / "detail.json"))
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["value"] > 0
    assert d["steps"] == int(extra[1]) and d["warmup"] == int(extra[3])
    if workload == "merkle":
        assert d["config"]["hashes_per_step"] == 2 * 65535 + 1
        assert d["combine_matches_recomputed"] is True
        assert d["dist"]["world_size"] == 2 and d["dist"]["backend"] == "gloo" and d["dist"]["ranks_reported"] == 2
        assert d["dist"]["per_rank_value"]["min"] <= d["dist"]["per_rank_value"]["max"]
        assert detail["value"] == pytest.approx(d["value"], rel=1e-8) and len(detail["dist"]["ranks"]) == 2
        # the AIR + FRI half of the metric at N > 1: independent 2^20-row jobs on every rank, n_gpus x the slowest rate
        a = d["airfri"]
        assert a["n_gpus"] == 2 and a["scaling"] == "weak" and a["pedersen_hashes_per_job"] == 6 * (1 << 22) - 128
        assert a["commits_per_sec"] == pytest.approx(2 * a["commits_per_sec_slowest_gpu"]) and a["commits_per_sec"] > 0
        assert d["summary"]["airfri_commits_per_sec"] == a["commits_per_sec"]
        assert list(d)[-1] == "summary"
    else:  # ONE 2^15- / 2^20-row proof over the two ranks: its roots are the single-GPU roots of the same trace
        assert d["config"]["rows_total"] == 2 << int(extra[5]) and d["sharded_roots_match_single_gpu"] is True
        assert d["config"]["exchange"]["per_fold"].startswith("none")


@pytest.mark.parametrize("workload,extra", [("merkle", ["--steps", "3", "--warmup", "1"]),
                                            ("airfri", ["--steps", "1", "--warmup", "0", "--log-rows", "15"]),
                                            # BASELINE configs[4] at its stated size: 2^24 rows in all, 8 ranks
                                            ("airfri", ["--steps", "1", "--warmup", "0", "--log-rows", "21"])])
def test_eight_ranks_share_one_gpu(workload, extra, tmp_path):
    """BASELINE configs[4] names EIGHT devices: the same launch with world = 8 (eight processes on device 0, gloo
    for the exchanges) - per-rank subtrees + all_gather + three top levels for the Merkle workload, and ONE
    2^18-row (and one 2^24-row: configs[4] at its stated size, 8 GiB through the all-to-all) trace as 16 coset units
    over eight ranks (two per rank, block-cyclic row shards, shard-local folds) whose roots must equal the single-GPU
    roots of the same trace."""
    env = dict(os.environ, STARKPERP_BENCH_SHARE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0", STARKPERP_WINDOW_BITS="16",
               STARKPERP_BENCH_DETAIL=str(tmp_path / "detail.json"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"),
           "--gpus", "8", "--workload", workload, "--window-bits", "0", "--no-extras", "--no-cpu-baseline",
           "--no-airfri"] + extra  # (the per-rank AIR + FRI jobs of the default line are covered by the two-rank test)
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=1500, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    d, _ = _line_and_detail(out, str(tmp_path / "detail.json"))
    assert d["n_gpus"] == 8 and d["scaling"] == "weak" and d["value"] > 0
    if workload == "merkle":
        assert d["config"]["hashes_per_step"] == 8 * 65535 + 7
        assert d["combine_matches_recomputed"] is True
        assert d["dist"]["world_size"] == 8 and d["dist"]["ranks_reported"] == 8
    else:
        assert d["config"]["rows_total"] == 8 << int(extra[5]) and d["sharded_roots_match_single_gpu"] is True
        assert d["config"]["exchange"]["per_fold"].startswith("none")


def test_gpus_2_without_a_launcher_spawns_its_own_ranks(tmp_path):
    """`python bench.py --gpus 2` with no RANK / WORLD_SIZE in the environment starts the two ranks
    itself (torch.distributed.run on 127.0.0.1) instead of exiting; rank 0's line carries what proves the process group
    (world size, backend, per-rank value, combine check - VERDICT r5 item 8), the detail file the full report (per-rank
    device / free HBM / window plan, the peer-access matrix) and the sustained-window fields."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(STARKPERP_BENCH_SHARE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0", STARKPERP_WINDOW_BITS="16",
               STARKPERP_BENCH_DETAIL=str(tmp_path / "detail.json"))
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "1",
           "--window-bits", "0", "--no-extras", "--no-cpu-baseline", "--no-airfri", "--min-timed-s", "0.3",
           "--preheat-s", "0.2"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    d, detail = _line_and_detail(out, str(tmp_path / "detail.json"))
    assert d["n_gpus"] == 2 and d["value"] > 0 and d["combine_matches_recomputed"] is True
    assert d["timed"]["total_s"] >= 0.3 and d["timed"]["preheat_s"] >= 0.2
    assert d["config"]["trees_in_timed_call"] == 4 and d["config"]["calls_per_region"] == 1
    assert d["dist"]["world_size"] == 2 and d["dist"]["ranks_reported"] == 2 and d["dist"]["peer_access_all"] is True
    assert d["dist"]["per_rank_value"]["min"] <= d["dist"]["per_rank_value"]["max"]
    assert len(d["build"]["lib_sha256_16"]) == 16 and detail["build"]["lib_sha256"].startswith(d["build"]["lib_sha256_16"])
    assert detail["burst"]["value"] > 0 and detail["sustained_over_burst"] > 0
    dist = detail["dist"]
    assert dist["world_size"] == 2 and len(dist["ranks"]) == 2
    assert [r["rank"] for r in dist["ranks"]] == [0, 1]
    assert all(r["window_bits"] == dist["ranks"][0]["window_bits"] for r in dist["ranks"])  # ONE table plan per job
    assert all(r["free_hbm_gib"] > 0 and r["local_hashes_per_sec"] > 0 for r in dist["ranks"])
    assert len(dist["peer_access"]) == 2
    assert "gfx950" in detail["build"]["compiled_with"]


def test_default_line_is_a_sustained_measurement(tmp_path):
    """VERDICT r4 item 1 + r5 item 1: the one-GPU line is ONE parseable line under 8 KB; it times at least
    --min-timed-s of regions after the pre-heat, carries the clock / power the device held in that window and the
    held-clock roofline fraction; the 50 ms burst and the telemetry windows are in the detail file."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "20", "--warmup", "5",
           "--window-bits", "0", "--no-extras", "--no-cpu-baseline", "--no-airfri", "--min-timed-s", "1.0",
           "--preheat-s", "0.5"]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env["STARKPERP_BENCH_DETAIL"] = str(tmp_path / "detail.json")
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    d, detail = _line_and_detail(out, str(tmp_path / "detail.json"))
    assert out.stdout.strip().count("\n") == 0  # nothing but the line on stdout
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "summary"):
        assert k in d, k
    assert list(d)[-1] == "summary" and d["cpu_baseline"] is None  # --no-cpu-baseline
    t = d["timed"]
    assert t["total_s"] >= 1.0 and t["preheat_s"] >= 0.5 and t["regions"] >= 100
    assert t["p10_s"] <= t["median_s"] <= t["p90_s"]
    assert d["value"] == pytest.approx(20 * 65535 / t["median_s"], rel=1e-6)
    tr = detail["timed_regions"]
    assert tr["min_s"] <= tr["median_s"] <= tr["max_s"] and tr["count"] == t["regions"]
    tel = detail["telemetry"]
    assert tel["source"] and tel["sustained"]["samples"] >= 10
    assert 500 < d["telemetry"]["sclk_mhz_median"] < 3000 and 100 < d["telemetry"]["power_w_median"] < 2000
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "avg_launch_us", "hbm", "whole_region", "sources"):
        assert k in r, k
    assert r["kernel"] == "ped_accumulate_kernel" and r["frac"] == pytest.approx(r["achieved"] / r["peak"], rel=1e-6)
    assert r["held_clock_mhz"] == d["telemetry"]["sclk_mhz_median"]
    assert r["frac_at_held_clock"] == pytest.approx(r["frac"] * 2400.0 / r["held_clock_mhz"], rel=1e-6)
    assert detail["burst"]["total_s"] < 0.2


def test_default_line_with_every_leg_fits_the_bound(tmp_path):
    """The driver's own command shape (every leg on: CPU baselines, airfri, extras), shortened windows: the line the
    driver will parse - under 8 KB, roofline and cpu_baseline objects present, parity bits true, C3 as a distribution."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "20", "--warmup", "5",
           "--window-bits", "0", "--min-timed-s", "0.5", "--preheat-s", "0.3"]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env["STARKPERP_BENCH_DETAIL"] = str(tmp_path / "detail.json")
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=1500, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    d, detail = _line_and_detail(out, str(tmp_path / "detail.json"))
    assert "failed_legs" not in d, (d.get("failed_legs"), out.stderr[-3000:])
    assert d["cpu_baseline"]["kind"] == "port" and d["cpu_baseline"]["value"] > 0 and d["cpu_baseline"]["matches_gpu"] is True
    assert d["roofline"]["frac"] > 0.3 and d["roofline"]["hbm"]["frac"] > 0
    assert d["airfri"]["commits_per_sec"] > 1 and d["airfri"]["cpu_baseline"]["commits_per_sec_scaled_to_2p20"] > 0
    s = d["summary"]
    assert all(s["parity_in_run"].values()), s["parity_in_run"]
    assert s["c3_one_call_ms"] > 0 and s["c3_one_call_ms_p90"] >= s["c3_one_call_ms"] >= s["c3_one_call_ms_min"]
    one = detail["extra"]["c3_4096_orders_one_call_seconds"]
    assert one["calls"] >= 20 and len(one["all"]) == one["calls"] and one["committed"] and one["all_verified"]

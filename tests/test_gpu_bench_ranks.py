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


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("workload,extra", [("merkle", ["--steps", "6", "--warmup", "2"]),
                                            ("airfri", ["--steps", "1", "--warmup", "1", "--log-rows", "14"]),
                                            ("airfri", ["--steps", "1", "--warmup", "0", "--log-rows", "19"])])
def test_two_ranks_share_one_gpu(workload, extra):
    env = dict(os.environ, STARKPERP_BENCH_SHARE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0",
               STARKPERP_WINDOW_BITS="16")  # two ranks share one GPU here: small tables whatever the caller set
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"),
           "--gpus", "2", "--workload", workload, "--window-bits", "0", "--no-extras", "--no-cpu-baseline"] + extra
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]  # rank 0 prints ONE line
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["value"] > 0
    assert d["steps"] == int(extra[1]) and d["warmup"] == int(extra[3])
    if workload == "merkle":
        assert d["config"]["hashes_per_step"] == 2 * 65535 + 1
        assert d["combine_matches_recomputed"] is True
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
def test_eight_ranks_share_one_gpu(workload, extra):
    """BASELINE configs[4] names EIGHT devices: the same launch with world = 8 (eight processes on device 0, gloo
    for the exchanges) - per-rank subtrees + all_gather + three top levels for the Merkle workload, and ONE
    2^18-row (and one 2^24-row: configs[4] at its stated size, 8 GiB through the all-to-all) trace as 16 coset units
    over eight ranks (two per rank, block-cyclic row shards, shard-local folds) whose roots must equal the single-GPU
    roots of the same trace."""
    env = dict(os.environ, STARKPERP_BENCH_SHARE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0", STARKPERP_WINDOW_BITS="16")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"),
           "--gpus", "8", "--workload", workload, "--window-bits", "0", "--no-extras", "--no-cpu-baseline",
           "--no-airfri"] + extra  # (the per-rank AIR + FRI jobs of the default line are covered by the two-rank test)
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=1500, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 8 and d["scaling"] == "weak" and d["value"] > 0
    if workload == "merkle":
        assert d["config"]["hashes_per_step"] == 8 * 65535 + 7
        assert d["combine_matches_recomputed"] is True
    else:
        assert d["config"]["rows_total"] == 8 << int(extra[5]) and d["sharded_roots_match_single_gpu"] is True
        assert d["config"]["exchange"]["per_fold"].startswith("none")


def test_gpus_2_without_a_launcher_spawns_its_own_ranks():
    """`python bench.py --gpus 2` with no RANK / WORLD_SIZE in the environment starts the two ranks
    itself (torch.distributed.run on 127.0.0.1) instead of exiting; rank 0's line carries the process-group report
    (per-rank device / free HBM / window plan, the peer-access matrix, per-rank value) and the sustained-window fields."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(STARKPERP_BENCH_SHARE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0", STARKPERP_WINDOW_BITS="16")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "1",
           "--window-bits", "0", "--no-extras", "--no-cpu-baseline", "--no-airfri", "--min-timed-s", "0.3",
           "--preheat-s", "0.2"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["value"] > 0 and d["combine_matches_recomputed"] is True
    assert d["timed_regions"]["total_s"] >= 0.3 and d["timed_regions"]["preheat_s"] >= 0.2
    assert d["burst"]["value"] > 0 and d["sustained_over_burst"] > 0
    assert d["config"]["trees_in_timed_call"] == 4 and d["config"]["calls_per_region"] == 1
    dist = d["dist"]
    assert dist["world_size"] == 2 and len(dist["ranks"]) == 2
    assert [r["rank"] for r in dist["ranks"]] == [0, 1]
    assert all(r["window_bits"] == dist["ranks"][0]["window_bits"] for r in dist["ranks"])  # ONE table plan per job
    assert all(r["free_hbm_gib"] > 0 and r["local_hashes_per_sec"] > 0 for r in dist["ranks"])
    assert dist["per_rank_value"]["min"] <= dist["per_rank_value"]["max"]
    assert len(dist["peer_access"]) == 2
    assert len(d["build"]["lib_sha256"]) == 64 and "gfx950" in d["build"]["compiled_with"]


def test_default_line_is_a_sustained_measurement(tmp_path):
    """the one-GPU line times at least --min-timed-s of regions after the pre-heat, carries the
    clock / power the device held in that window and the held-clock roofline fraction, and keeps the 50 ms burst."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "20", "--warmup", "5",
           "--window-bits", "0", "--no-extras", "--no-cpu-baseline", "--no-airfri", "--min-timed-s", "1.0",
           "--preheat-s", "0.5"]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][0])
    tr = d["timed_regions"]
    assert tr["total_s"] >= 1.0 and tr["preheat_s"] >= 0.5 and tr["count"] >= 100
    assert tr["min_s"] <= tr["median_s"] <= tr["max_s"]
    assert d["value"] == pytest.approx(20 * 65535 / tr["median_s"])
    tel = d["telemetry"]
    assert tel["source"] and tel["sustained"]["samples"] >= 10
    assert 500 < tel["sclk_mhz_median"] < 3000 and 100 < tel["power_w_median"] < 2000
    r = d["roofline"]
    assert r["held_clock_mhz"] == tel["sclk_mhz_median"]
    assert r["frac_at_held_clock"] == pytest.approx(r["frac"] * 2400.0 / tel["sclk_mhz_median"], rel=1e-6)
    assert d["burst"]["total_s"] < 0.2

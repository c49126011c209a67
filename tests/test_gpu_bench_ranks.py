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

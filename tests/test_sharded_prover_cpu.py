"""One AIR+FRI commit job sharded over 2, 4 and 8 gloo ranks (starkperp.sharded_prover: LDE units with one
interpolation per column -> ONE all-to-all into block-cyclic row shards with halos -> per-block subtrees +
all_gather of block roots -> shard-LOCAL folds -> one all_gather -> replicated tail) against the same job
computed in one process by the oracle.  The stage kernels are replaced by the oracle (tests/oracle_ops.py);
the layout and exchange logic is the product's.  8 ranks is the world size BASELINE.json configs[4] names."""
import os
import random

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import oracle_ops as O

P = O.P


def _worker(rank, world, port, n_hashes, log_block, q, final_log=6):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from starkperp import sharded_prover as SP
    rng = random.Random(91)
    inputs = [(rng.randrange(P), rng.randrange(P)) for _ in range(n_hashes)]
    alphas = [rng.randrange(P) for _ in range(O.S.N_CONSTRAINTS)]
    log_lde = (512 * n_hashes).bit_length() - 1 + 2
    betas = [rng.randrange(P) for _ in range(log_lde - final_log)]
    trace, want_roots, want_final = O.single_process_job(inputs, alphas, betas, final_log) if rank == 0 else (None, None, None)
    trace = O.S.pedersen_trace(inputs) if trace is None else trace
    cols = torch.stack([O.to_tensor(c) for c in trace])
    stats = {}
    roots, final = SP.commit_job(O.OracleOps(), dist, cols, alphas, betas, final_log=final_log, log_block=log_block,
                                stats=stats)
    # every rank must end with the same roots; rank 0 also holds the single-process answer
    gathered = [None] * world
    dist.all_gather_object(gathered, (roots, final, stats))
    ok = all(g[:2] == gathered[0][:2] for g in gathered)
    if rank == 0:
        ok = ok and roots == want_roots and final == want_final
        # the only bulk traffic is the LDE all-to-all: (world - 1) / world of 4 columns x 4 n felts + halos,
        # plus block roots and ONE tail gather - no per-fold exchange
        n = 512 * n_hashes
        B = 1 << stats["log_block"]
        lde_bytes = 4 * 4 * n * 32
        total_sent = sum(g[2]["bytes_sent_by_this_rank"] for g in gathered)
        bound = lde_bytes * (world - 1) // world * (B + 4) // B + world * (world - 1) * (160 * (4 * n // B // world) + B * 32)
        if total_sent > bound:
            ok = "traffic %d > %d" % (total_sent, bound)
        # the cosets a rank owns share their column's interpolation
        if not all(g[2]["interpolations"] == -(-g[2]["units_owned"] // 4) for g in gathered):
            ok = "interpolations %r" % [(g[2]["interpolations"], g[2]["units_owned"]) for g in gathered]
    q.put((rank, ok, len(roots)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,n_hashes,log_block", [(2, 2, 6), (4, 4, 7), (2, 4, 11), (8, 2, 4), (8, 4, 8)])
def test_sharded_job_equals_single_process_job(world, n_hashes, log_block):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29700 + (os.getpid() % 2000) + 7 * world + n_hashes
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_hashes, log_block, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=600) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    n_roots = 2 + ((512 * n_hashes).bit_length() - 1 + 2 - 7)
    assert sorted(results) == [(r, True, n_roots) for r in range(world)]


def test_final_layer_that_is_still_sharded_is_gathered():
    """final_log chosen so that 2^final_log >= world * B: the fold loop ends while the layer is block-cyclic; every
    rank must still return the WHOLE final layer in natural order, equal to the single-process one."""
    world, n_hashes, log_block, final_log = 2, 2, 4, 7   # world * B = 32 positions, final layer 128
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29700 + (os.getpid() % 2000) + 977
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_hashes, log_block, q, final_log)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=600) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    n_roots = 2 + ((512 * n_hashes).bit_length() - 1 + 2 - final_log - 1)
    assert sorted(results) == [(r, True, n_roots) for r in range(world)]


def test_single_rank_path_equals_single_process_job():
    """world = 1 (dist = None) goes through the same code: units, shard assembly, halo, folds."""
    from starkperp import sharded_prover as SP
    rng = random.Random(92)
    inputs = [(rng.randrange(P), rng.randrange(P))]
    alphas = [rng.randrange(P) for _ in range(O.S.N_CONSTRAINTS)]
    betas = [rng.randrange(P) for _ in range(5)]
    trace, want_roots, want_final = O.single_process_job(inputs, alphas, betas)
    cols = torch.stack([O.to_tensor(c) for c in trace])
    roots, final = SP.commit_job(O.OracleOps(), None, cols, alphas, betas)
    assert roots == want_roots and final == want_final

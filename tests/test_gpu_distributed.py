"""SURVEY 8(e) on the library's own kernels with the world size BASELINE names: EIGHT ranks (eight processes sharing
the one GPU of the test box, gloo for the 8 x 32-byte exchanges) - the height-64 multi-update sharded by key prefix
and the sharded rebuild, against one tree in one process and against the oracle's from-scratch roots."""
import os
import random

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import ref_py as R

pytestmark = pytest.mark.gpu
WORLD = 8


def _worker(rank, port, height, batches, leaves, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), STARKPERP_WINDOW_BITS="16",
                      HSA_ENABLE_IPC_MODE_LEGACY="0")
    dist.init_process_group("gloo", rank=rank, world_size=WORLD)
    from starkperp import batch, distributed as D, state
    tree = D.ShardedSparseTree(dist, torch, height, lambda h, empty: state.LibrarySparseTree(h, empty),
                               batch.pedersen_hash_many)
    got = [tree.update(mods) for mods in batches]
    lo, hi = D.shard_range(len(leaves), rank, WORLD)
    root = D.sharded_merkle_root(dist, torch, leaves[lo:hi], batch.merkle_root, batch.pedersen_hash_many)
    q.put((rank, got, root))
    dist.barrier()
    dist.destroy_process_group()


def test_eight_ranks_multi_update_and_rebuild_on_the_gpu():
    rng = random.Random(91)
    height = 64
    batches = [{rng.randrange(2**height): rng.randrange(1, 2**64) for _ in range(40)} for _ in range(3)]
    batches.append({0: 5, 2**height - 1: 6, 2**61: 7, 2**61 - 1: 8})  # both ends of the key space and a shard boundary
    leaves = [rng.randrange(R.FIELD_PRIME) for _ in range(1 << 12)]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 32500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, port, height, batches, leaves, q)) for r in range(WORLD)]
    for p in procs:
        p.start()
    results = sorted(q.get(timeout=600) for _ in range(WORLD))
    for p in procs:
        p.join(timeout=120)
    # one tree in one process (the library's persistent tree) ...
    from starkperp import state
    single = state.LibrarySparseTree(height, 0)
    expect = [single.update(m) for m in batches]
    single.close()
    assert all(got == expect for _, got, _ in results)
    # ... and the oracle's own walk from scratch over everything written (no tree code of the product involved)
    written = {}
    for m in batches:
        written.update(m)
    assert expect[-1][1] == R.merkle_multi_update_sparse(height, written)
    want_root = R.merkle_root(leaves)
    assert all(root == want_root for _, _, root in results)

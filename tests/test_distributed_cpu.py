"""world_size-2 gloo test of the N>1 path (shard -> all_gather of sub-roots -> top of tree).
The exchange logic is the product's (starkperp.distributed); the hash plugged in is the oracle,
so this runs on CPU and checks the sharded root equals the single-process root."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import workloads as wl
from oracle import ref_py as R


def _worker(rank, world, port, leaves, expect, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from starkperp import distributed as D
    lo, hi = D.shard_range(len(leaves), rank, world)
    hash_many = lambda a, b: [R.pedersen_hash(x, y) for x, y in zip(a, b)]
    root = D.sharded_merkle_root(dist, torch, leaves[lo:hi], R.merkle_root, hash_many)
    q.put((rank, root == expect))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4, 8])
def test_sharded_root_matches_single(world):
    leaves = wl.leaves(16, seed=55)
    expect = R.merkle_root(leaves)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000) + world
    procs = [ctx.Process(target=_worker, args=(r, world, port, leaves, expect, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    assert sorted(results) == [(r, True) for r in range(world)]


def _forest_worker(rank, world, port, trees, expect, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from starkperp import distributed as D
    hash_many = lambda a, b: [R.pedersen_hash(x, y) for x, y in zip(a, b)]
    per = len(trees[0]) // world
    local = [R.merkle_root(t[rank * per : (rank + 1) * per]) for t in trees]
    roots = D.combine_forest_subroots(dist, torch, local, hash_many)
    q.put((rank, roots == expect))
    dist.barrier()
    dist.destroy_process_group()


def test_forest_combine_matches_single():
    """nb = 4 trees sharded over 2 ranks: the tree-major regrouping of the gathered sub-roots
    (the same layout bench.py's device path uses) reproduces every tree's root."""
    world = 2
    trees = [wl.leaves(8, seed=70 + i) for i in range(4)]
    expect = [R.merkle_root(t) for t in trees]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_forest_worker, args=(r, world, port, trees, expect, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    assert sorted(results) == [(r, True) for r in range(world)]
    from starkperp import distributed as D
    assert D.forest_gather_layout(["r0t0", "r0t1", "r1t0", "r1t1"], 2, 2) == ["r0t0", "r1t0", "r0t1", "r1t1"]


def test_shard_range_covers():
    from starkperp import distributed as D
    for n in (0, 1, 7, 8, 4096):
        for world in (1, 2, 3, 8):
            spans = [D.shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))


def _sharded_tree_worker(rank, world, port, height, batches, expect, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from starkperp import distributed as D
    from starkperp import state
    hash_many = lambda a, b: [R.pedersen_hash(x, y) for x, y in zip(a, b)]
    make_tree = lambda h, empty: state.SparseMerkleTree(h, empty, hash_many=hash_many)
    tree = D.ShardedSparseTree(dist, torch, height, make_tree, hash_many)
    got = [tree.update(mods) for mods in batches]
    q.put((rank, got == expect))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4, 8])
def test_sharded_multi_update_matches_single_tree(world):
    """Multi-update sharded by key prefix: every rank owns the subtree of its top key
    bits, sub-roots are all-gathered, the top levels are hashed everywhere - same (old, new) root
    pairs as one tree in one process, batch after batch (oracle hash, gloo)."""
    import random
    from starkperp import state
    height = 10
    rng = random.Random(3)
    batches = []
    for _ in range(3):
        batches.append({rng.randrange(2**height): rng.randrange(1, R.FIELD_PRIME) for _ in range(6)})
    batches.append({0: 5, 2**height - 1: 6})
    hash_many = lambda a, b: [R.pedersen_hash(x, y) for x, y in zip(a, b)]
    single = state.SparseMerkleTree(height, 0, hash_many=hash_many)
    expect = [single.update(m) for m in batches]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + (os.getpid() % 2000) + world
    procs = [ctx.Process(target=_sharded_tree_worker, args=(r, world, port, height, batches, expect, q))
             for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=300) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    assert sorted(results) == [(r, True) for r in range(world)]

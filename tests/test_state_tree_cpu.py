"""Host bookkeeping of the persistent sparse Merkle tree (starkperp.state.SparseMerkleTree) with
the ORACLE hash plugged in: checks the induced-subtree walk, sibling reuse and the facts store
against a from-scratch recomputation.  CPU only (the GPU run of the same class is in
tests/test_gpu_state.py)."""
import random

from oracle import ref_py as R
from starkperp.state import SparseMerkleTree

P = R.FIELD_PRIME


def oracle_hash_many(xs, ys):
    return [R.pedersen_hash(a, b) for a, b in zip(xs, ys)]


def test_updates_match_from_scratch_roots():
    rng = random.Random(8)
    height = 6
    tree = SparseMerkleTree(height, 0, hash_many=oracle_hash_many)
    assert tree.root == R.empty_subtree_roots(height)[height]
    state = {}
    for round_ in range(3):
        mods = {rng.randrange(1 << height): rng.randrange(P) for _ in range(5)}
        old, new = tree.update(mods)
        assert old == R.merkle_multi_update_sparse(height, state)
        state.update(mods)
        assert new == R.merkle_multi_update_sparse(height, state) == tree.root
        for k, v in state.items():
            assert tree.get(k) == v
    leaves = [state.get(i, 0) for i in range(1 << height)]
    assert tree.root == R.merkle_root(leaves)
    assert tree.update({}) == (tree.root, tree.root)
    # overwriting with the same value leaves the root unchanged
    k = next(iter(state))
    assert tree.update({k: state[k]})[1] == tree.root


def test_height_64_two_far_keys():
    tree = SparseMerkleTree(64, 0, hash_many=oracle_hash_many)
    mods = {0: 9, 2**64 - 1: 7}
    assert tree.update(mods)[1] == R.merkle_multi_update_sparse(64, mods)
    assert tree.get(2**64 - 1) == 7 and tree.get(12345) == 0

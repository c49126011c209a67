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


def test_shared_state_updates_with_oracle_hash():
    """SharedState bookkeeping (squash, prev/new position hashing, dual-tree update) with the oracle
    hash plugged in, against a from-scratch recomputation."""
    import pytest
    from starkperp.state import SharedState, squash_updates
    assert squash_updates([(5, "a", "b"), (2, "x", "y"), (5, "b", "c")]) == [(2, "x", "y"), (5, "a", "c")]
    with pytest.raises(AssertionError):
        squash_updates([(5, "a", "b"), (5, "zzz", "c")])
    ph = lambda ps: [R.position_hash(p[0], p[1], list(p[2])) for p in ps]
    st = SharedState(8, 6, hash_many=oracle_hash_many, position_hashes=ph)
    empty = (0, 0, ())
    assert st.positions_root == R.empty_subtree_roots(8, R.position_hash(0, 0, []))[8]
    p1 = (123, 50, ((7, 1, -2),))
    p2 = (123, 40, ((7, 1, 3),))
    q1 = (456, -9, ())
    (old_p, new_p), (old_o, new_o) = st.apply_state_updates(
        [(3, empty, p1), (200, empty, q1), (3, p1, p2)], [(9, 0, 10), (9, 10, 25), (1, 0, 4)])
    leaves = {3: R.position_hash(p2[0], p2[1], list(p2[2])), 200: R.position_hash(q1[0], q1[1], [])}
    assert new_p == R.merkle_multi_update_sparse(8, leaves, R.position_hash(0, 0, []))
    assert new_o == R.merkle_multi_update_sparse(6, {9: 25, 1: 4}) and old_o == R.empty_subtree_roots(6)[6]
    # a second batch continues from the stored state; a stale previous value is rejected
    st.apply_state_updates([(200, q1, q1)], [(1, 4, 6)])
    assert st.orders_root == R.merkle_multi_update_sparse(6, {9: 25, 1: 6})
    with pytest.raises(AssertionError):
        st.apply_state_updates([(3, p1, p1)], [])


def test_apply_state_updates_is_all_or_nothing():
    """state/state.cairo:135-186 fails the WHOLE batch: a bad orders list (inconsistent accesses, a stale
    previous value, an out-of-range leaf) must leave the positions root where it was, and a failure
    inside the second tree update rolls the first one back."""
    import pytest
    from starkperp.state import SharedState
    ph = lambda ps: [R.position_hash(p[0], p[1], list(p[2])) for p in ps]
    st = SharedState(8, 6, hash_many=oracle_hash_many, position_hashes=ph)
    empty = (0, 0, ())
    p1 = (123, 50, ((7, 1, -2),))
    st.apply_state_updates([(3, empty, p1)], [(9, 0, 10)])
    roots = (st.positions_root, st.orders_root)
    p2 = (123, 40, ((7, 1, 3),))
    for bad_orders in ([(9, 10, 11), (9, 99, 12)],          # inconsistent chain of accesses
                       [(9, 7, 11)],                         # previous value is not what the tree holds
                       [(9, 10, R.FIELD_PRIME)],             # unhashable-range leaf
                       [(1 << 6, 0, 1)]):                    # key outside the tree
        with pytest.raises(AssertionError):
            st.apply_state_updates([(3, p1, p2)], bad_orders)
        assert (st.positions_root, st.orders_root) == roots
    # a failure raised by the second update itself: the first update is undone
    real_update = st.orders.update

    def failing_update(mods):
        raise AssertionError("Unhashable input.")
    st.orders.update = failing_update
    with pytest.raises(AssertionError):
        st.apply_state_updates([(3, p1, p2)], [(9, 10, 11)])
    st.orders.update = real_update
    assert (st.positions_root, st.orders_root) == roots
    st.apply_state_updates([(3, p1, p2)], [(9, 10, 11)])
    assert st.positions_root != roots[0] and st.orders_root != roots[1]

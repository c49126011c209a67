"""GPU parity of the callers around the hash kernels: position leaves, sparse multi-update,
and the 4096-order batch of BASELINE.json configs[2] against the reference-generated golden."""
import json
import os
import random

import pytest

import workloads as wl
from oracle import ref_py as R

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
P, N = R.FIELD_PRIME, R.EC_ORDER


def load(name):
    return json.load(open(os.path.join(GOLD, name)))


def h(s):
    return int(s, 16)


def test_position_hashes():
    from starkperp import state
    g = load("g6_merkle.json")
    poss = wl.positions(64, seed=3)
    assert state.position_hashes_many(poss) == [h(v) for v in g["position_hashes_seed3"]]
    assert state.position_hash((0, 0, [])) == h(g["empty_position_leaf"])


def test_sparse_roots():
    from starkperp import batch
    g = load("g6_merkle.json")
    mods = {k: h(v) for k, v in g["sparse_h20_seed77"]["mods"]}
    assert batch.merkle_sparse_root(20, mods) == h(g["sparse_h20_seed77"]["root"])
    emp = [h(v) for v in g["empty_roots_leaf0"]]
    for hgt in (0, 1, 5, 64):
        assert batch.merkle_sparse_root(hgt, {}) == emp[hgt]
    # dense case equals the full rebuild; ragged key patterns vs the oracle
    lv = wl.leaves(32, seed=9)
    assert batch.merkle_sparse_root(5, dict(enumerate(lv))) == R.merkle_root(lv)
    rng = random.Random(4)
    for hgt, cnt in ((3, 3), (7, 9), (10, 1)):
        mods = {rng.randrange(1 << hgt): rng.randrange(P) for _ in range(cnt)}
        assert batch.merkle_sparse_root(hgt, mods, 5) == R.merkle_multi_update_sparse(hgt, mods, 5)
    assert batch.merkle_sparse_root(64, {2**64 - 1: 7, 0: 9}) == R.merkle_multi_update_sparse(
        64, {2**64 - 1: 7, 0: 9})


def test_c3_order_batch():
    """4096 limit orders: message hashes, key derivation, signing, verification and the height-64
    orders-tree update, all against the golden produced by the reference."""
    from starkperp import batch, perpetual_messages as pm, state
    g = load("g7_c3_batch.json")
    orders = wl.limit_orders(g["n"], seed=g["orders_seed"])
    zs = pm.limit_order_msgs_many([wl.order_args(o) for o in orders])
    assert wl.digest_felts(zs) == g["z_digest"]
    assert [hex(v) for v in zs[:4]] == g["z_first4"]
    keys = wl.private_keys(1024, seed=g["keys_seed"])
    pubs = batch.public_keys_many(keys)
    assert wl.digest_felts([q[0] for q in pubs]) == g["pub_digest"]
    zsig = [z % 2**251 for z in zs]
    sigs = batch.sign_many(zsig, [keys[o["key_index"]] for o in orders])
    assert wl.digest_felts([r for r, _ in sigs]) == g["r_digest"]
    assert wl.digest_felts([s for _, s in sigs]) == g["s_digest"]
    rs = [r for r, _ in sigs]
    ss = [(s % (N - 1)) + 1 if i % 16 == 5 else s for i, (_, s) in enumerate(sigs)]
    ok = batch.verify_many(zsig, rs, ss, [pubs[o["key_index"]][0] for o in orders])
    assert "".join("1" if v else "0" for v in ok) == g["verify_bits"]
    mods = {state.order_id_of(z): o["amount_synthetic"] for z, o in zip(zs, orders)}
    assert len(mods) == g["n_distinct_order_ids"]
    assert state.orders_tree_root(mods, g["orders_tree_height"]) == h(g["orders_tree_root"])


def test_persistent_tree_and_position_updates():
    """SparseMerkleTree on the GPU backend: same roots as the oracle-hashed twin, old/new root
    pairs of successive multi-updates in a height-64 tree; hash_position_updates (hash.cairo:76-131)."""
    from starkperp import state
    rng = random.Random(31)
    gpu = state.SparseMerkleTree(64)
    ref_state = {}
    for _ in range(3):
        mods = {rng.randrange(2**64): rng.randrange(P) for _ in range(8)}
        old, new = gpu.update(mods)
        assert old == R.merkle_multi_update_sparse(64, ref_state)
        ref_state.update(mods)
        assert new == R.merkle_multi_update_sparse(64, ref_state)
    some = next(iter(ref_state))
    assert gpu.get(some) == ref_state[some]
    poss = wl.positions(8, seed=5)
    changed = [(p[0], p[1] + 1, p[2]) for p in poss]
    ups = [(i, poss[i], changed[i] if i % 2 else poss[i]) for i in range(8)]
    got = state.hash_position_updates(ups)
    for (k, prev, new), (gk, gp, gn) in zip(ups, got):
        assert gk == k and gp == R.position_hash(*prev) and gn == R.position_hash(*new)


def test_combine_forest_dev_with_fake_collective():
    """bench.py's N > 1 combine on one GPU: a stand-in `dist` whose all_gather returns this rank's
    sub-roots from every 'rank' (deterministically perturbed), checked against the oracle."""
    import torch
    from starkperp import _lib, stark
    from starkperp.distributed import combine_forest_dev

    lib = _lib.ensure_init()
    for world in (2, 4):
        for nb in (1, 3, 4):
            rng = random.Random(world * 10 + nb)
            per_rank = [[rng.randrange(P) for _ in range(nb)] for _ in range(world)]  # [rank][tree]

            class FakeDist:
                @staticmethod
                def get_world_size(group=None):
                    return world

                @staticmethod
                def all_gather_into_tensor(out, inp):
                    flat = [v for r in range(world) for v in per_rank[r]]
                    out.copy_(stark.felts_to_tensor(flat))

            roots = stark.felts_to_tensor(per_rank[0])
            gathered = torch.zeros((world * nb, 4), dtype=torch.int64, device="cuda")
            top = torch.zeros((nb * (2 * world - 1), 4), dtype=torch.int64, device="cuda")
            out = combine_forest_dev(lib, FakeDist, roots, gathered, top, nb,
                                     torch.cuda.current_stream().cuda_stream)
            torch.cuda.synchronize()
            got = stark.tensor_to_felts(out)
            exp = [R.merkle_root([per_rank[r][t] for r in range(world)]) for t in range(nb)]
            assert got == exp, (world, nb)


def test_shared_state_on_gpu_matches_oracle_twin():
    from starkperp.state import SharedState
    ph = lambda ps: [R.position_hash(p[0], p[1], list(p[2])) for p in ps]
    oracle_hash_many = lambda a, b: [R.pedersen_hash(x, y) for x, y in zip(a, b)]
    gpu, ref = SharedState(64, 64), SharedState(64, 64, hash_many=oracle_hash_many, position_hashes=ph)
    assert gpu.positions_root == ref.positions_root and gpu.orders_root == ref.orders_root
    empty = (0, 0, ())
    poss = wl.positions(6, seed=9)
    accesses = [(1000 + 77 * i, empty, tuple([p[0], p[1], tuple(p[2])])) for i, p in enumerate(poss)]
    orders = [(2**63 + i, 0, 5 * i + 1) for i in range(5)]
    assert gpu.apply_state_updates(accesses, orders) == ref.apply_state_updates(accesses, orders)
    # ... and not only its twin: the roots recomputed FROM SCRATCH by the oracle's own sparse-tree walk
    # (oracle/ref_py.py merkle_multi_update_sparse, no SharedState / squash code involved)
    leaves = {k: R.position_hash(p[0], p[1], list(p[2])) for k, _, p in accesses}
    assert gpu.positions_root == R.merkle_multi_update_sparse(64, leaves, R.position_hash(0, 0, []))
    assert gpu.orders_root == R.merkle_multi_update_sparse(64, {k: v for k, _, v in orders})


def test_library_tree_matches_the_python_tree_over_many_batches():
    """sp_tree_* (the library keeps the tree) against SparseMerkleTree with the oracle's hash over a
    sequence of batches: fresh keys, overwritten keys, neighbours that share parents, keys reset
    to the empty leaf, a single-leaf update, an empty update, heights 3, 16 and 64."""
    import random
    from oracle import cref
    from starkperp import state

    def oracle_hash_many(xs, ys):
        return cref.pedersen_hash_many(list(xs), list(ys))[0]

    rng = random.Random(11)
    for height, empty_leaf, rounds, per_round in ((3, 0, 6, 3), (16, 5, 5, 40), (64, 0, 4, 150)):
        lib_tree = state.LibrarySparseTree(height, empty_leaf)
        ref_tree = state.SparseMerkleTree(height, empty_leaf, hash_many=oracle_hash_many)
        assert lib_tree.root == ref_tree.root
        known = []
        for r in range(rounds):
            mods = {}
            for _ in range(per_round):
                choice = rng.random()
                if known and choice < 0.3:
                    k = rng.choice(known)                      # overwrite
                elif known and choice < 0.5:
                    k = rng.choice(known) ^ 1                  # sibling of an existing leaf
                else:
                    k = rng.randrange(2**height)
                mods[k] = empty_leaf if rng.random() < 0.1 else rng.randrange(P)
            if r == 2:
                mods = {rng.randrange(2**height): 7}           # single leaf
            if r == 3:
                mods = {}                                      # nothing
            known += list(mods)
            assert lib_tree.update(mods) == ref_tree.update(mods), (height, r)
            assert lib_tree.root == ref_tree.root
            probe = rng.sample(known, min(len(known), 10)) + [rng.randrange(2**height)]
            assert lib_tree.get_many(probe) == ref_tree.get_many(probe)
        lib_tree.close()


def test_library_tree_rejects_bad_input_and_keeps_its_state():
    from starkperp import _lib, state
    t = state.LibrarySparseTree(8, 0)
    before = t.update({3: 9, 200: 11})[1]
    with pytest.raises(AssertionError):
        t.update({256: 1})                                     # key out of range (Python-side check)
    import ctypes
    lib = _lib.ensure_init()
    keys = (ctypes.c_uint64 * 2)(5, 5)                          # not strictly increasing
    old, new, st = _lib.new_felts(1), _lib.new_felts(1), _lib.new_bytes(1)
    assert lib.sp_tree_update(t._handle, keys, _lib.pack_felts([1, 2]), 2, old, new, st) < 0
    keys = (ctypes.c_uint64 * 1)(5)
    rc = lib.sp_tree_update(t._handle, keys, _lib.pack_felts([P]), 1, old, new, st)  # leaf >= p
    assert rc == 0 and st[0] == 1 and _lib.unpack_felts(new, 1)[0] == before
    assert t.root == before and t.get(5) == 0
    assert lib.sp_tree_update(12345, keys, _lib.pack_felts([1]), 1, old, new, st) < 0  # unknown handle
    t.close()


def test_library_tree_large_batches_and_table_growth():
    """Device-resident tree at the sizes of BASELINE configs[2] and beyond: batches of 3000-5000 leaves at
    height 64 (several 1024-node tiles per level in the structure kernel, the hash table growing and
    rehashing twice), dense runs of adjacent keys (pairs that are both new), overwrites, then a failed
    update in the middle - against the Python tree with the optimised C comparator's hash."""
    import random
    from oracle import cref
    from starkperp import state

    def oracle_hash_many(xs, ys):
        return cref.opt_pedersen_hash_many(list(xs), list(ys))[0]

    rng = random.Random(23)
    lib_tree = state.LibrarySparseTree(64, 0)
    ref_tree = state.SparseMerkleTree(64, 0, hash_many=oracle_hash_many)
    known = []
    for r, n in enumerate((3000, 5000, 4096, 700)):
        mods = {}
        base = rng.randrange(2**63)
        for i in range(n):
            c = rng.random()
            if known and c < 0.2:
                k = rng.choice(known)
            elif c < 0.5:
                k = base + i                                   # a dense run: siblings inside the batch
            else:
                k = rng.randrange(2**64)
            mods[k] = rng.randrange(P)
        known += list(mods)
        assert lib_tree.update(mods) == ref_tree.update(mods), r
        probe = rng.sample(known, 50) + [rng.randrange(2**64)]
        assert lib_tree.get_many(probe) == ref_tree.get_many(probe)
        if r == 1:  # an update that fails leaves every level untouched
            root = lib_tree.root
            import numpy as np
            from starkperp import batch_np as bn
            bad = {**{rng.randrange(2**64): rng.randrange(P) for _ in range(2000)}, 12345: P}  # one leaf == p
            with pytest.raises(AssertionError):  # detected by the hash kernel on the device, 64 levels later
                lib_tree.update_arrays(np.array(list(bad), dtype=np.uint64), bn.felts_from_ints(list(bad.values())))
            assert lib_tree.root == root and lib_tree.get_many(probe) == ref_tree.get_many(probe)
    lib_tree.close()


def test_library_tree_grows_through_three_tables_without_losing_a_node():
    """Round 6 growth policy (csrc/merkle.hip tree_reserve): a table that must grow is sized for four times the need,
    the rehash is enqueued on the tree's stream and the OLD table is retired, not freed, until the next growth.  48
    small batches take a fresh tree from its first 2^16-slot table through two growths (the second one frees the table
    retired by the first).  Checked: every old root handed back is the previous new root; the final root equals the
    root of ONE from-scratch multi-update of everything written - on the device (sp_merkle_sparse_root, a different
    code path) and in the Python tree with the C comparator's hash; keys written before either growth are still there."""
    import random
    from oracle import cref
    from starkperp import state

    def oracle_hash_many(xs, ys):
        return cref.opt_pedersen_hash_many(list(xs), list(ys))[0]

    rng = random.Random(61)
    lib_tree = state.LibrarySparseTree(64, 0)
    first = {rng.randrange(2**64): rng.randrange(1, P) for _ in range(50)}
    _, root = lib_tree.update(first)
    written = dict(first)
    for r in range(47):
        mods = {rng.randrange(2**64): rng.randrange(1, P) for _ in range(50)}
        if r % 5 == 0:  # overwrite some early keys as well: replaced nodes must not be counted twice
            for k in rng.sample(list(first), 5):
                mods[k] = rng.randrange(1, P)
        written.update(mods)
        old, new = lib_tree.update(mods)
        assert old == root and new != root, r
        root = new
    keys = sorted(written)
    assert lib_tree.root == root and lib_tree.get_many(keys) == [written[k] for k in keys]
    assert root == state.orders_tree_root(written, 64)  # one from-scratch multi-update on the device
    ref_tree = state.SparseMerkleTree(64, 0, hash_many=oracle_hash_many)
    assert ref_tree.update(written)[1] == root            # and on the CPU: 64 level calls instead of 48 x 64
    lib_tree.close()


def test_order_batch_in_one_call_matches_the_separate_calls():
    """sp_order_batch (BASELINE configs[2] in one library call): message hashes, verdicts and the orders-tree roots
    equal those of the separate entry points; a bad signature leaves the tree uncommitted; equal order ids are an
    error; concurrent updates of two trees from two threads do not disturb each other."""
    import threading
    import numpy as np
    import workloads as wl
    from starkperp import batch, batch_np as bn, perpetual_messages as pm, state
    orders = wl.limit_orders(512, seed=21)
    keys = wl.private_keys(64, seed=22)
    pubs = batch.public_keys_many(keys)
    args = [wl.order_args(o) for o in orders]
    words = np.stack([bn.felts_from_ints(col) for col in zip(*[pm._limit_order_words(*a) for a in args])])
    zs = pm.limit_order_msgs_many(args)
    sigs = batch.sign_many([z % 2**251 for z in zs], [keys[o["key_index"] % 64] for o in orders])
    r, s = bn.felts_from_ints([a for a, _ in sigs]), bn.felts_from_ints([b for _, b in sigs])
    qx = bn.felts_from_ints([pubs[o["key_index"] % 64][0] for o in orders])
    leaves = bn.felts_from_ints([o["amount_synthetic"] + 1 for o in orders])
    tree, twin = state.LibrarySparseTree(64, 0), state.LibrarySparseTree(64, 0)
    seedling = {5: 7, 2**63 + 11: 9}
    tree.update(seedling), twin.update(seedling)
    z, verdicts, old, new, committed = bn.order_batch(words, r, s, qx, tree, leaves)
    assert committed and bn.ints_from_felts(z) == zs and verdicts.tolist() == [1] * 512
    want_old, want_new = twin.update({state.order_id_of(zz): o["amount_synthetic"] + 1 for zz, o in zip(zs, orders)})
    assert (old, new) == (want_old, want_new) and tree.root == twin.root
    # one bad signature: verdict 0, nothing committed
    s_bad = s.copy()
    s_bad[17, 0] ^= np.uint64(2)
    leaves2 = bn.felts_from_ints([o["amount_synthetic"] + 2 for o in orders])
    z2, verdicts2, old2, new2, committed2 = bn.order_batch(words, r, s_bad, qx, tree, leaves2)
    assert not committed2 and verdicts2[17] == 0 and int(verdicts2.sum()) == 511 and old2 == new2 == want_new
    assert tree.root == want_new and tree.get(state.order_id_of(zs[3])) == orders[3]["amount_synthetic"] + 1
    # two orders with the same id
    dup = np.concatenate([words[:, :4], words[:, :1]], axis=1)
    pick = [0, 1, 2, 3, 0]
    with pytest.raises(Exception):
        bn.order_batch(dup, r[pick], s[pick], qx[pick], tree, leaves[pick])
    assert tree.root == want_new
    # two trees updated from two threads at once (each on its own stream, no library lock across the levels)
    rng = random.Random(4)
    batches = [{rng.randrange(2**64): rng.randrange(1, 2**64) for _ in range(2048)} for _ in range(4)]
    got = {}

    def run(name, t):
        got[name] = [t.update(b) for b in batches]
    ts = [threading.Thread(target=run, args=(nm, t)) for nm, t in (("a", tree), ("b", twin))]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert got["a"] == got["b"] and tree.root == twin.root
    tree.close(), twin.close()


def test_order_batch_error_paths_release_the_verifier():
    """Round 6: sp_order_batch parks its verifier thread at a gate until the tree update has enqueued its levels.
    Every way out of the call must open that gate and join the thread - a path that forgets would hang the caller.
    Each case runs under a watchdog: an unknown tree handle (the tree fails before it enqueues anything), two orders
    with one id (refused after the verifier was spawned), a leaf out of range (detected on the device, 64 levels
    later, nothing committed), and a good batch afterwards on the same tree."""
    import threading
    import numpy as np
    import workloads as wl
    from starkperp import batch, batch_np as bn, perpetual_messages as pm, state
    orders = wl.limit_orders(256, seed=31)
    keys = wl.private_keys(32, seed=32)
    pubs = batch.public_keys_many(keys)
    args = [wl.order_args(o) for o in orders]
    words = np.stack([bn.felts_from_ints(col) for col in zip(*[pm._limit_order_words(*a) for a in args])])
    zs = pm.limit_order_msgs_many(args)
    sigs = batch.sign_many([z % 2**251 for z in zs], [keys[o["key_index"] % 32] for o in orders])
    r, s = bn.felts_from_ints([a for a, _ in sigs]), bn.felts_from_ints([b for _, b in sigs])
    qx = bn.felts_from_ints([pubs[o["key_index"] % 32][0] for o in orders])
    leaves = bn.felts_from_ints([o["amount_synthetic"] + 1 for o in orders])
    tree = state.LibrarySparseTree(64, 0)
    tree.update({9: 9})
    root0 = tree.root

    def guarded(fn):
        box = {}

        def run():
            try:
                box["value"] = fn()
            except BaseException as e:  # noqa: BLE001 - the outcome is inspected by the caller
                box["error"] = e
        t = threading.Thread(target=run, daemon=True)
        t.start()
        t.join(120)
        assert not t.is_alive(), "sp_order_batch did not return: the verifier was left at its gate"
        return box

    class Ghost:  # a handle the library never issued
        _handle = 987654
    out = guarded(lambda: bn.order_batch(words, r, s, qx, Ghost(), leaves))
    assert "error" in out and tree.root == root0
    dup = np.concatenate([words[:, :4], words[:, :1]], axis=1)
    pick = [0, 1, 2, 3, 0]
    out = guarded(lambda: bn.order_batch(dup, r[pick], s[pick], qx[pick], tree, leaves[pick]))
    assert "error" in out and tree.root == root0
    bad_leaves = leaves.copy()
    bad_leaves[5] = bn.felts_from_ints([P])[0]  # leaf == p: out of range, found by the hash kernel
    out = guarded(lambda: bn.order_batch(words, r, s, qx, tree, bad_leaves))
    assert "error" in out and tree.root == root0
    out = guarded(lambda: bn.order_batch(words, r, s, qx, tree, leaves))
    z, verdicts, old, new, committed = out["value"]
    assert committed and old == root0 and new == tree.root != root0 and verdicts.tolist() == [1] * 256
    tree.close()


def test_order_batch_refuses_a_message_hash_of_2p251_or_more():
    """order.cairo:22 / constants.cairo:57: a message hash is a signed message only below 2^251 (the order id is its
    top 64 bits); signature.py:227 asserts the same bound.  With depth 1 the words ARE the message hashes, so the
    case can be stated: the order with z >= 2^251 gets SP_VERIFY_ASSERT_MSG, the others their verdicts, and the tree
    stays as it was - the hash is NOT reduced modulo 2^251 and verified as another message."""
    import numpy as np
    import workloads as wl
    from starkperp import batch, batch_np as bn, state
    keys = wl.private_keys(8, seed=31)
    pubs = batch.public_keys_many(keys)
    rng = random.Random(32)
    zs = [rng.randrange(2**251) for _ in range(8)]
    zs[3] = rng.randrange(2**192)  # so that 2^251 + z is still a field element
    sigs = batch.sign_many(zs, keys)
    over = 2**251 + zs[3]  # < p; the same low 251 bits as the message order 3 was signed for
    assert over < 2**251 + 17 * 2**192 + 1
    words = bn.felts_from_ints(zs[:3] + [over] + zs[4:])[None]
    r, s = bn.felts_from_ints([a for a, _ in sigs]), bn.felts_from_ints([b for _, b in sigs])
    qx = bn.felts_from_ints([p[0] for p in pubs])
    leaves = bn.felts_from_ints(list(range(1, 9)))
    tree = state.LibrarySparseTree(64, 0)
    before = tree.update({7: 9})[1]
    z, verdicts, old, new, committed = bn.order_batch(words, r, s, qx, tree, leaves)
    assert not committed and old == new == before == tree.root
    assert verdicts.tolist() == [1, 1, 1, 5, 1, 1, 1, 1] and bn.ints_from_felts(z)[3] == over
    # the same orders with the signed message itself: committed
    words_ok = bn.felts_from_ints(zs)[None]
    z, verdicts, old, new, committed = bn.order_batch(words_ok, r, s, qx, tree, leaves)
    assert committed and verdicts.tolist() == [1] * 8 and old == before and new == tree.root != before
    assert tree.get(zs[3] >> 187) == 4
    tree.close()

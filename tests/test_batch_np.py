"""NumPy batch entry points (starkperp.batch_np): word packing without big integers (CPU), and on the GPU
the same results as the list API / the reference-generated goldens."""
import json
import os
import random

import numpy as np
import pytest

import workloads as wl
from oracle import ref_py as R

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
P = R.FIELD_PRIME


def order_arrays(orders):
    from starkperp import batch_np as bn
    sell, buy, fee, a_sell, a_buy = [], [], [], [], []
    for o in orders:
        syn, col, buying, f, a_syn, a_col, a_fee, nonce, pos, exp = wl.order_args(o)
        s, b, ns, nb = (col, syn, a_col, a_syn) if buying else (syn, col, a_syn, a_col)
        sell.append(s); buy.append(b); fee.append(f); a_sell.append(ns); a_buy.append(nb)
    args = [wl.order_args(o) for o in orders]
    u = lambda i: np.array([a[i] for a in args], dtype=np.uint64)
    return (bn.felts_from_ints(sell), bn.felts_from_ints(buy), bn.felts_from_ints(fee),
            np.array(a_sell, dtype=np.uint64), np.array(a_buy, dtype=np.uint64), u(6), u(7), u(8), u(9))


def test_limit_order_words_match_the_integer_packer():
    from starkperp import batch_np as bn
    from starkperp import perpetual_messages as pm
    orders = wl.limit_orders(200, seed=5)
    words = bn.limit_order_words(*order_arrays(orders))
    assert words.shape == (5, 200, 4)
    for i, o in enumerate(orders):
        assert [bn.ints_from_felts(words[k, i : i + 1])[0] for k in range(5)] == pm._limit_order_words(*wl.order_args(o))
    rng = random.Random(3)
    vals = [rng.randrange(2**251) for _ in range(50)] + [0, 2**251 - 1, 2**187, 2**187 - 1]
    assert list(bn.order_ids(bn.felts_from_ints(vals))) == [v >> 187 for v in vals]
    assert bn.ints_from_felts(bn.felts_from_ints(vals)) == vals
    # maximal fields do not bleed into their neighbours
    full = bn.pack_fields(1, [(2**64 - 1, 0), (2**64 - 1, 64), (2**64 - 1, 130), (3, 250)])
    assert bn.ints_from_felts(full)[0] == (2**64 - 1) | ((2**64 - 1) << 64) | ((2**64 - 1) << 130) | (3 << 250)


@pytest.mark.gpu
def test_numpy_entry_points_equal_the_list_api():
    from starkperp import batch, batch_np as bn, state
    g = json.load(open(os.path.join(GOLD, "g5_messages.json")))
    orders = wl.limit_orders(256, seed=g["seed"])
    z = bn.limit_order_msgs(*order_arrays(orders))
    assert bn.ints_from_felts(z) == [int(v, 16) for v in g["limit_order_z"]]
    rng = random.Random(9)
    xs, ys = [rng.randrange(P) for _ in range(300)], [rng.randrange(P) for _ in range(300)]
    assert bn.ints_from_felts(bn.pedersen_hash_many(bn.felts_from_ints(xs), bn.felts_from_ints(ys))) == \
        batch.pedersen_hash_many(xs, ys)
    with pytest.raises(AssertionError):
        bn.pedersen_hash_many(bn.felts_from_ints([P]), bn.felts_from_ints([1]))
    # signatures: valid, corrupted, and a pre-assert
    keys = [rng.randrange(1, R.EC_ORDER) for _ in range(32)]
    pubs = batch.public_keys_many(keys)
    zs = [rng.randrange(2**251) for _ in range(32)]
    sigs = batch.sign_many(zs, keys)
    rs, ss = [r for r, _ in sigs], [s if i % 3 else s ^ 1 for i, (_, s) in enumerate(sigs)]
    qx = [q[0] for q in pubs]
    want = batch.verify_codes(zs, rs, ss, qx)
    got = bn.verify_codes(*(bn.felts_from_ints(v) for v in (zs, rs, ss, qx)))
    assert list(got) == want and set(want) == {0, 1}
    assert list(bn.verify_many(*(bn.felts_from_ints(v) for v in (zs, rs, ss, qx)))) == [c == 1 for c in want]
    with pytest.raises(AssertionError):
        bn.verify_many(*(bn.felts_from_ints(v) for v in ([1], [1], [0], qx[:1])))
    # tree update from arrays == from a dict
    a, b = state.LibrarySparseTree(64, 0), state.LibrarySparseTree(64, 0)
    mods = {rng.randrange(2**64): rng.randrange(P) for _ in range(500)}
    ks = list(mods)
    assert a.update(mods) == b.update_arrays(np.array(ks, dtype=np.uint64), bn.felts_from_ints([mods[k] for k in ks]))
    assert a.root == b.root
    a.close(); b.close()

"""Multi-asset order message (services/exchange/cairo/signature_message_hashes.cairo:387-471).
CPU part: the host packer with the oracle's hash injected equals the oracle's own restatement, the
word layout is checked field by field on a hand-built order, and the reference's fixture is recorded
for what it pins (the signature of its message hash) and what it does not (the packing, see
starkperp/exchange_messages.py).  GPU part: batched chains equal the oracle."""
import json
import os
import random

import pytest

from oracle import ref_py as R

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
P = R.FIELD_PRIME


def random_order(rng, n_give, n_receive, n_conditions, n_third):
    signer = rng.randrange(1, 2**251)
    give = [(rng.randrange(2**64), signer, rng.randrange(2**250), rng.randrange(2**64)) for _ in range(n_give)]
    receive = []
    for i in range(n_receive):
        key = rng.randrange(1, 2**251) if i < n_third else signer
        receive.append((rng.randrange(2**64), key, rng.randrange(2**250), rng.randrange(2**64)))
    rng.shuffle(receive)
    conditions = [rng.randrange(2**250) for _ in range(n_conditions)]
    return (signer, rng.randrange(2**32), rng.randrange(2**32), rng.randrange(2**126), give, receive, conditions)


def test_word_layout_field_by_field():
    from starkperp import exchange_messages as em
    signer, other = 0x111, 0x222
    give = [(1, signer, 0xA1, 10), (2, signer, 0xA2, 20)]
    receive = [(3, other, 0xB1, 30), (4, signer, 0xB2, 40), (5, other, 0xB3, 50)]
    words = em.multi_asset_order_words(signer, 7, 9, 0x5, give, receive, [0xC1])
    assert words[0] == 0xC1                                    # conditions first
    assert words[1:6] == [0xB1, 0xB2, 0xB3, 0xA1, 0xA2]        # assets: receive, then give
    assert words[6:8] == [other, other]                        # third-party keys
    fields = [3, 30, 4, 40, 5, 50, 1, 10, 2, 20]
    assert words[8] == (fields[0] << 128) + (fields[1] << 64) + fields[2]
    assert words[9] == (fields[3] << 128) + (fields[4] << 64) + fields[5]
    assert words[10] == (fields[6] << 128) + (fields[7] << 64) + fields[8]
    assert words[11] == fields[9]                              # a lone trailing field is not shifted
    assert words[12] == (0 << 12) + 2                          # indices 0 and 2 of `receive`
    meta = words[13]
    assert meta & 7 == 0 and meta < 2**251
    meta >>= 3
    assert meta & (2**126 - 1) == 0x5
    meta >>= 126
    assert [(meta >> s) & 0xFFF for s in (36, 24, 12, 0)] == [2, 3, 2, 1]  # n_give, n_receive, n_third, n_cond
    assert (meta >> 48) & 0xFFFFFFFF == 9 and (meta >> 80) & 0xFFFFFFFF == 7 and meta >> 112 == 6
    assert len(words) == 14


def test_host_packer_equals_oracle_restatement():
    from starkperp import exchange_messages as em
    rng = random.Random(77)
    for shape in [(1, 1, 0, 0), (3, 2, 2, 1), (2, 25, 1, 23), (4, 4, 3, 4)]:
        o = random_order(rng, *shape)
        assert em.multi_asset_order_words(*o) == R.multi_asset_order_words(*o)
    o = random_order(rng, 2, 2, 1, 1)
    assert em.multi_asset_order_hash(*o, hash_function=R.pedersen_hash) == R.multi_asset_order_hash(*o)


def test_reference_fixture_signature_is_pinned_and_its_packing_is_not():
    """signature_test_data.json:102-139,185-188 (copied as data into reference_kats.json): the message
    hash carries the deterministic signature of the fixture's key - that part follows from the
    reference.  The hash does not follow from the fixture's fields under the Cairo source of this
    tree; the test records the value the restated format gives so that a change is noticed."""
    fx = json.load(open(os.path.join(GOLD, "reference_kats.json")))["multi_asset_order"]
    priv, z = int(fx["private_key"], 16), int(fx["message_hash"], 16)
    key = R.private_to_stark_key(priv)
    r, s = int(fx["signature"]["r"], 16), int(fx["signature"]["s"], 16)
    assert R.sign(z, priv) == (r, s) and R.verify(z, r, s, key)
    assert int(fx["receive"][0]["public_key"], 16) == key  # the first reception is the signer's own vault

    def info(e):
        return (int(e["vault_id"]), int(e.get("public_key", hex(key)), 16), int(e["asset_id"], 16), int(e["amount"]))
    got = R.multi_asset_order_hash(key, fx["nonce"], fx["expiration_timestamp"], int(fx["system_id"], 16),
                                   [info(e) for e in fx["give"]], [info(e) for e in fx["receive"]],
                                   [int(c, 16) for c in fx["conditions"]])
    assert got == 0x63dcfb5d90eb12bca3545706ddbaf05b9ac90f6960233ea0fbcc39b0964dce1
    assert got != z


@pytest.mark.xfail(strict=True, reason="the reference's only multi_asset_order vector (signature_test_data.json:185-188) "
                                      "does not follow from signature_message_hashes.cairo:387-471 of this tree, nor from "
                                      "107 520 layout variants around it (tools/search_multi_asset_layout.py, "
                                      "profiles/r03_multi_asset_layout_search.txt): the packing is UNPINNED.  strict: the "
                                      "day a restatement reproduces the fixture this test passes and the suite says so")
def test_reference_fixture_hash_is_reproduced():
    fx = json.load(open(os.path.join(GOLD, "reference_kats.json")))["multi_asset_order"]
    key = int(fx["receive"][0]["public_key"], 16)

    def info(e):
        return (int(e["vault_id"]), int(e.get("public_key", hex(key)), 16), int(e["asset_id"], 16), int(e["amount"]))
    got = R.multi_asset_order_hash(key, fx["nonce"], fx["expiration_timestamp"], int(fx["system_id"], 16),
                                   [info(e) for e in fx["give"]], [info(e) for e in fx["receive"]],
                                   [int(c, 16) for c in fx["conditions"]])
    assert got == int(fx["message_hash"], 16)


@pytest.mark.gpu
def test_multi_asset_orders_on_gpu_match_oracle():
    from starkperp import exchange_messages as em
    rng = random.Random(78)
    orders = [random_order(rng, *shape) for shape in
              [(1, 1, 0, 0), (3, 2, 2, 1), (2, 25, 1, 23), (4, 4, 3, 4), (3, 2, 2, 1), (1, 1, 0, 0), (2, 3, 0, 2)]]
    got = em.multi_asset_order_msgs_many(orders)
    assert got == [R.multi_asset_order_hash(*o) for o in orders]
    assert em.multi_asset_order_hash(*orders[1]) == got[1]  # scalar path through the GPU hash

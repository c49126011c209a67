"""Pins the oracle (oracle/ref_py.py) against the reference: the reference's own vectors
(tests/golden/reference_kats.json, copied data) and vectors produced by importing the reference in
the build container (oracle/gen_golden.py).  CPU only."""
import hashlib
import json
import os

import pytest

import workloads as wl
from oracle import ref_py as R

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    return json.load(open(os.path.join(GOLD, name)))


def h(s):
    return int(s, 16)


def test_params_and_table_digest():
    g = load("params_digest.json")
    assert h(g["FIELD_PRIME"]) == R.FIELD_PRIME and h(g["EC_ORDER"]) == R.EC_ORDER
    assert g["ALPHA"] == R.ALPHA and h(g["BETA"]) == R.BETA and g["FIELD_GEN"] == R.FIELD_GEN
    assert g["n_points"] == len(R.CONSTANT_POINTS) == 506
    assert g["constant_points_sha256"] == R.constant_points_digest()
    for i, (x, y) in g["sample_points"].items():
        assert R.CONSTANT_POINTS[int(i)] == [h(x), h(y)]


def test_constants_rederived_from_the_digits_of_pi():
    """nothing_up_my_sleeve_gen.py:50-91 restated: beta and all 506 points follow from pi alone and
    equal the table whose digest the reference's pedersen_params.json gave."""
    import hashlib
    beta, table = R.generate_constant_points(6)
    assert beta == R.BETA
    m = hashlib.sha256()
    for x, y in table:
        m.update(x.to_bytes(32, "big") + y.to_bytes(32, "big"))
    assert m.hexdigest() == load("params_digest.json")["constant_points_sha256"]
    assert R.pi_digits(40) == "3141592653589793238462643383279502884197"


def test_reference_hash_and_key_kats():
    k = load("reference_kats.json")
    for case in k["hash_test"].values():
        assert R.pedersen_hash(h(case["input_1"]), h(case["input_2"])) == h(case["output"])
    for priv, pub in list(k["keys_precomputed"].items()):
        assert R.private_to_stark_key(h(priv)) == h(pub)
    c = k["stark_cli_hash"]
    assert R.pedersen_hash(h(c["x"]), h(c["y"])) == h(c["out"])


def test_reference_signature_kat():
    a = load("reference_kats.json")["party_a_order"]
    z, d = h(a["message_hash"]), h(a["private_key"])
    r, s = R.sign(z, d)
    assert (r, s) == (h(a["signature"]["r"]), h(a["signature"]["s"]))
    assert R.private_to_stark_key(d) == h(a["public_key"])
    assert R.verify(z, r, s, h(a["public_key"]))


def test_reference_signature_fixtures():
    for name, f in load("reference_kats.json")["signature_fixtures"].items():
        got = R.verify(h(f["message_hash"]), h(f["r"]), h(f["s"]), h(f["public_key"]))
        assert got == f["reference_verify"], name


def test_rfc6979_a25():
    q = 0xFFFFFFFF00000000FFFFFFFFFFFFFFFFBCE6FAADA7179E84F3B9CAC2FC632551
    x = 0xC9AFA9D845BA75166B5C215767B1D6934E50C3DB36E89B127B8A622B120F6721
    k = R._rfc6979_k(q, x, hashlib.sha256(b"sample").digest())
    assert k == 0xA6E3C57DD01ABE90086538398355DD4C3B17AA873382B0F24D6129493D8AAD60


def test_reference_message_kats():
    m = load("reference_kats.json")["perpetual_messages"]
    for exp, d in m["limit_order"].items():
        got = R.get_limit_order_msg(
            d["assetIdSynthetic"], d["assetIdCollateral"], d["isBuyingSynthetic"], d["assetIdFee"],
            d["amountSynthetic"], d["amountCollateral"], d["amountFee"], d["nonce"],
            d["positionId"], d["expirationTimestamp"])
        assert hex(got) == exp
    for exp, d in m["transfer"].items():
        got = R.get_transfer_msg(
            d["assetId"], d["assetIdFee"], d["receiverPublicKey"], d["senderPositionId"],
            d["receiverPositionId"], d["feePositionId"], d["nonce"], d["amount"],
            d["maxAmountFee"], d["expirationTimestamp"])
        assert hex(got) == exp
    for exp, d in m["conditional_transfer"].items():
        got = R.get_conditional_transfer_msg(
            d["assetId"], d["assetIdFee"], d["receiverPublicKey"], d["condition"],
            d["senderPositionId"], d["receiverPositionId"], d["srcFeePositionId"], d["nonce"],
            d["amount"], d["maxAmountFee"], d["expirationTimestamp"])
        assert hex(got) == exp
    for exp, d in m["withdrawal_to_address"].items():
        got = R.get_withdrawal_to_address_msg(
            d["assetIdCollateral"], d["positionId"], d["ethAddress"], d["nonce"],
            d["expirationTimestamp"], d["amount"])
        assert hex(got) == exp
        # the Cairo program's type-7 branch puts the owner key where the eth address is (withdrawal.cairo:61-64)
        assert hex(R.withdrawal_hash(d["assetIdCollateral"], d["positionId"], int(d["ethAddress"], 16),
                                     12345, d["nonce"], d["expirationTimestamp"], d["amount"])) == exp
    assert m["withdrawal"], "the reference's type-6 withdrawal vector is part of the fixture"
    for exp, d in m["withdrawal"].items():  # perpetual_messages_precomputed.json:16-24
        got = R.get_withdrawal_msg(d["assetIdCollateral"], d["positionId"], d["nonce"],
                                   d["expirationTimestamp"], d["amount"])
        assert hex(got) == exp
        assert hex(R.withdrawal_hash(d["assetIdCollateral"], d["positionId"], 777, 777, d["nonce"],
                                     d["expirationTimestamp"], d["amount"])) == exp


def test_g1_pedersen_sample_and_edges():
    g = load("g1_pedersen.json")
    pairs = wl.pedersen_pairs(g["n"], seed=g["seed"])
    assert wl.digest_felts(h(v) for v in g["all"]) == g["digest"]
    for i in list(range(0, 1024, 16)):
        assert R.pedersen_hash(*pairs[i]) == h(g["all"][i])
    for a, b, o in g["edge"]:
        assert R.pedersen_hash(h(a), h(b)) == h(o)
    ar = g["arity"]
    assert R.pedersen_hash() == h(ar["zero"]) == R.SHIFT_POINT[0]
    assert R.pedersen_hash(1) == h(ar["one_1"])
    assert R.pedersen_hash(R.FIELD_PRIME - 1) == h(ar["one_pm1"])
    assert list(R.pedersen_hash_as_point(1, 2)) == [h(v) for v in ar["point_1_2"]]
    with pytest.raises(AssertionError):
        R.pedersen_hash(R.FIELD_PRIME, 0)
    with pytest.raises(AssertionError):
        R.pedersen_hash(1, 2, 3)


def test_g2_keys_sample():
    for d, x, y in load("g2_keys.json")["keys"][::8]:
        assert R.private_key_to_ec_point_on_stark_curve(h(d)) == (h(x), h(y))


def test_g3_sign_sample():
    cases = load("g3_sign.json")["cases"]
    for z, d, sd, r, s, k in cases[:40] + cases[40::12]:
        seed = None if sd is None else h(sd)
        assert R.generate_k_rfc6979(h(z), h(d), seed) == h(k)
        assert R.sign(h(z), h(d), seed) == (h(r), h(s))


def _key(c):
    return tuple(h(v) for v in c["key"]) if isinstance(c["key"], list) else h(c["key"])


def test_g4_verify_cases():
    cases = load("g4_verify.json")["cases"]
    picked = cases[:12] + cases[96:]
    for c in picked:
        try:
            got = "true" if R.verify(h(c["z"]), h(c["r"]), h(c["s"]), _key(c)) else "false"
        except AssertionError as e:
            msg = str(e)
            got = "assert:" + (msg.split(" ")[0] if msg else "")
        assert got == c["expect"], c["label"]


def test_g5_messages():
    g = load("g5_messages.json")
    orders = wl.limit_orders(256, seed=g["seed"])
    for o, z in list(zip(orders, g["limit_order_z"]))[::16]:
        assert R.get_limit_order_msg(*wl.order_args(o)) == h(z)
    assert R.get_transfer_msg(5, 6, 7, 8, 9, 10, 11, 12, 13, 14) == h(g["transfer"])
    assert R.get_conditional_transfer_msg(5, 6, 7, 99, 8, 9, 10, 11, 12, 13, 14) == h(
        g["conditional_transfer"])
    assert R.get_withdrawal_to_address_msg(5, 6, "0xabcdef0123", 7, 8, 9) == h(
        g["withdrawal_to_address"])
    assert R.get_price_msg(0x4D616B6572, 0x42544355534400000000000000000000, 0x5F590C1E,
                           0xAC9F3163AD52B000) == h(g["price"])


def test_g6_merkle_small():
    g = load("g6_merkle.json")
    for hgt in range(0, 6):
        lv = wl.leaves(1 << hgt, seed=100 + hgt)
        assert R.merkle_root(lv) == h(g["roots_seed_100_plus_h"][str(hgt)])
    emp = R.empty_subtree_roots(8)
    assert emp == [h(v) for v in g["empty_roots_leaf0"][:9]]
    poss = wl.positions(64, seed=3)
    for pos, exp in list(zip(poss, g["position_hashes_seed3"]))[::8]:
        assert R.position_hash(*pos) == h(exp)
    assert R.position_hash(0, 0, []) == h(g["empty_position_leaf"])


def test_extra_reference_fixtures():
    """g8: every signature fixture of signature_test_data.json with the reference's verdict, and
    the reference's deterministic signatures for every (message_hash, private_key) it holds."""
    g = load("g8_reference_fixtures_extra.json")
    for name, c in g["verify"].items():
        assert R.verify(h(c["message_hash"]), h(c["r"]), h(c["s"]), h(c["public_key"])) == c["reference_verify"], name
    for name, c in g["sign"].items():
        assert R.sign(h(c["message_hash"]), h(c["private_key"])) == (h(c["r"]), h(c["s"])), name
        assert R.private_to_stark_key(h(c["private_key"])) == h(c["public_key"]), name


def test_grind_key_published_vectors():
    """The reference's own key-grinding known answers (key_derivation.spec.js:45-52 "Key grinding" and
    :63-70, where the seed is the r half of an Ethereum signature); signature.py:263-288 is the
    Python twin of that routine."""
    from starkperp import signature as S
    kats = {
        0x86F3E7293141F20A8BAFF320E8EE4ACCB9D4A4BF2B4D295E8CEE784DB46E0519:
            0x5C8C8683596C732541A59E03007B2D30DBBBB873556FE65B5FB63C16688F941,
        0x21FBF0696D5E0AA2EF41A2B4FFB623BCAF070461D61CF7251C74161F82FEC3A4:
            0x766F11E90CD7C7B43085B56DA35C781F8C067AC0D578EABDCEEBC4886435BDA,
    }
    for seed, want in kats.items():
        assert R.grind_key(seed, R.EC_ORDER) == want
        assert S.grind_key(seed, S.EC_ORDER) == want

"""Host-side pieces of the message layer that need no GPU: Keccak-256 (for build_condition,
reference perpetual_messages.py:15-21) and the word packers, checked with the oracle's Pedersen
hash injected through the `hash_function=` seam."""
import hashlib
import json
import os
import random

from oracle import ref_py as R
from starkperp import keccak

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_keccak256_published_vectors():
    assert keccak.keccak256(b"").hex() == "c5d2460186f7233c927e7db2dcc703c0e500b653ca82273b7bfad8045d85a470"
    assert keccak.keccak256(b"abc").hex() == "4e03657aea45a94fc7d47ba826c8d667c0d1e6e33a64a036ec44f58fa12d6c45"


def test_sponge_matches_hashlib_sha3_for_every_block_boundary():
    rng = random.Random(5)
    for n in list(range(0, 140)) + [271, 272, 273, 1000]:
        m = bytes(rng.randrange(256) for _ in range(n))
        assert keccak.sponge256(m, 0x06) == hashlib.sha3_256(m).digest(), n


def test_build_condition_is_the_masked_keccak_of_the_packed_pair():
    from starkperp import perpetual_messages as pm
    address = "0x" + "12" * 20
    fact = bytes(range(32))
    want = int.from_bytes(keccak.keccak256(bytes.fromhex("12" * 20) + fact), "big") % 2**250
    assert pm.build_condition(address, fact) == want
    assert pm.build_condition(address[2:], fact) == want
    assert 0 <= want < R.FIELD_PRIME
    for bad_address, bad_fact in [("0x1234", fact), (address, b"short")]:
        try:
            pm.build_condition(bad_address, bad_fact)
        except ValueError:
            continue
        raise AssertionError("accepted a malformed condition input")


def test_scalar_builders_with_the_oracle_hash_match_the_reference_kats():
    """The four precomputed message files of the reference (perpetual_messages_test.py), through
    our word packers with the oracle hash injected; no GPU involved."""
    from starkperp import perpetual_messages as pm
    m = json.load(open(os.path.join(GOLD, "reference_kats.json")))["perpetual_messages"]
    H = R.pedersen_hash
    for exp, d in m["limit_order"].items():
        assert hex(pm.get_limit_order_msg(
            d["assetIdSynthetic"], d["assetIdCollateral"], d["isBuyingSynthetic"], d["assetIdFee"],
            d["amountSynthetic"], d["amountCollateral"], d["amountFee"], d["nonce"], d["positionId"],
            d["expirationTimestamp"], hash_function=H)) == exp
    for exp, d in m["withdrawal_to_address"].items():
        assert hex(pm.get_withdrawal_to_address_msg(
            d["assetIdCollateral"], d["positionId"], d["ethAddress"], d["nonce"],
            d["expirationTimestamp"], d["amount"], hash_function=H)) == exp
    for exp, d in m["withdrawal"].items():  # type 6, withdrawal.cairo:57-60
        assert hex(pm.get_withdrawal_msg(
            d["assetIdCollateral"], d["positionId"], d["nonce"], d["expirationTimestamp"], d["amount"],
            hash_function=H)) == exp
        assert hex(pm.withdrawal_hash(d["assetIdCollateral"], d["positionId"], 5, 5, d["nonce"],
                                      d["expirationTimestamp"], d["amount"], hash_function=H)) == exp
    # both branches of the Cairo selector against the oracle's restatement, and the bounds
    for owner, signer in ((9, 9), (9, 10)):
        assert pm.withdrawal_hash(3, 4, owner, signer, 5, 6, 7, hash_function=H) == \
            R.withdrawal_hash(3, 4, owner, signer, 5, 6, 7)
    for bad in ((2**250, 1, 1, 1, 1), (1, 2**64, 1, 1, 1), (1, 1, 2**32, 1, 1), (1, 1, 1, 2**32, 1),
                (1, 1, 1, 1, 2**64), (1, 1, -1, 1, 1)):
        try:
            pm.get_withdrawal_msg(*bad, hash_function=H)
        except AssertionError:
            continue
        raise AssertionError("accepted an out-of-range withdrawal field")


def test_pi_as_string_leading_digits():
    from starkperp.math_utils import pi_as_string
    for d in (2, 10, 77, 632):  # 632 = 76 * 7 + 100, the reference's request (nothing_up_my_sleeve_gen.py:57)
        ours = pi_as_string(d)
        assert ours[: max(1, d - 2)] == R.pi_digits(d)[: max(1, d - 2)]
    try:
        import mpmath
    except ImportError:
        return
    saved = mpmath.mp.dps
    try:
        mpmath.mp.dps = 632
        assert pi_as_string(632)[:600] == ("3" + str(mpmath.mp.pi)[2:])[:600]
    finally:
        mpmath.mp.dps = saved


def test_device_nonce_model_equals_the_host_generator():
    """csrc/rfc6979.hpp relies on two simplifications of signature.py:117-134 + python-ecdsa's
    generate_k for this curve: h1 is the message hash itself (the one-nibble pad and bits2int cancel)
    and a candidate is int(V) >> 4.  The same simplified procedure in Python must equal the host
    generator (starkperp/rfc6979.py, pinned by the reference's signatures) on every input class."""
    import hashlib
    import hmac
    from starkperp.signature import EC_ORDER, generate_k_rfc6979

    def model(z, d, seed):
        material = d.to_bytes(32, "big") + z.to_bytes(32, "big")
        if seed:
            material += seed.to_bytes((seed.bit_length() + 7) // 8, "big")
        prf = lambda key, msg: hmac.new(key, msg, hashlib.sha256).digest()
        v, k = b"\x01" * 32, b"\x00" * 32
        k = prf(k, v + b"\x00" + material)
        v = prf(k, v)
        k = prf(k, v + b"\x01" + material)
        v = prf(k, v)
        while True:
            v = prf(k, v)
            cand = int.from_bytes(v, "big") >> 4
            if 1 <= cand < EC_ORDER:
                return cand
            k = prf(k, v + b"\x00")
            v = prf(k, v)

    rng = random.Random(1)
    for bits in [1, 2, 8, 9, 100, 243, 244, 245, 247, 248, 249, 250, 251]:
        for _ in range(6):
            z = rng.randrange(2 ** (bits - 1), 2**bits) if bits > 1 else rng.randrange(2)
            d = rng.randrange(1, EC_ORDER)
            for seed in (None, 0, 1, 255, 256, 2**32, 2**64 - 1):
                assert model(z, d, seed) == generate_k_rfc6979(z, d, seed), (bits, seed)

"""SURVEY 8(f) N4 on the GPU: the range-check builtin's encoding (eight 16-bit limbs, sorted pool, permutation
product in a second committed phase) and ONE trace holding the Pedersen, ECDSA and range-check builtin segments
of a batch - kernels against the oracle (oracle/stark_ref.py "rc16"), proofs against the CPU verifier
(verify_builtins_proof), tampered proofs rejected.  Build-defined, parity unpinned (the reference has no
prover); every commitment hash is the pinned pedersen_hash."""
import copy
import random

import pytest

from oracle import cref
from oracle import ref_py as R
from oracle import stark_ref as S

pytestmark = pytest.mark.gpu
P = S.P


def c_hash(a, b):
    return cref.pedersen_hash_many([a], [b])[0][0]


@pytest.fixture(scope="module")
def stark():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from starkperp import stark as st
    return st


def narrow_values(rng, count, lo=300, hi=420):
    """range-checked values whose limbs lie in a narrow band, so that a short trace can fill the holes"""
    return [sum(rng.randrange(lo, hi) << (16 * k) for k in range(8)) for _ in range(count)]


def test_rc16_kernels_match_oracle(stark):
    rng = random.Random(61)
    values = narrow_values(rng, 100)
    padded, lo, hi = stark.rc16_fill(values, 128)
    assert (padded, lo, hi) == S.rc16_fill(values, 128)
    want = S.rc16_trace(padded)
    cols = stark.rc16_columns(stark.felts_to_tensor(padded))
    n = cols.shape[1]
    assert n == 1024 and [stark.tensor_to_felts(c) for c in cols] == want
    z = rng.randrange(P)
    p = stark.rc16_product(cols[0], cols[2], z)
    want_p = S.rc16_product_column(want[0], want[2], z)
    assert stark.tensor_to_felts(p) == want_p and want_p[-1] == 1
    alphas = [rng.randrange(P) for _ in range(8)]
    cols_lde, p_lde = stark.lde(cols), stark.lde(p.unsqueeze(0))[0]
    per = stark.periodic_lde(n, 3, "cuda", "rc16")
    comp = stark.air_eval_rc16(cols_lde, p_lde, per, n, alphas, z, lo, hi)
    want_comp = S.rc16_composition_on_coset([S.lde(c) for c in want], S.lde(want_p), n, alphas, z, lo, hi)
    assert stark.tensor_to_felts(comp) == want_comp
    assert S.poly_degree_bound_check(want_comp, S.GEN, 2 * n)
    # the prefix product over a length that needs all three scan passes (2^15 cells)
    big = narrow_values(rng, 4096, 0, 512)
    padded, lo, hi = stark.rc16_fill(big, 4096)
    cols = stark.rc16_columns(stark.felts_to_tensor(padded))
    p = stark.tensor_to_felts(stark.rc16_product(cols[0], cols[2], z))
    tr = S.rc16_trace(padded)
    assert p == S.rc16_product_column(tr[0], tr[2], z)


def small_batch(rng, n_hashes, n_sigs, n_values):
    from starkperp import batch
    hashes = [(rng.randrange(P), rng.randrange(P)) for _ in range(n_hashes)]
    keys = [rng.randrange(1, R.EC_ORDER) for _ in range(n_sigs)]
    zs = [rng.randrange(1, 2**251) for _ in range(n_sigs)]
    sigs = batch.sign_many(zs, keys)
    pubs = batch.public_keys_many(keys)
    signatures = [(z, r, s, q) for z, (r, s), q in zip(zs, sigs, pubs)]
    return hashes, signatures, narrow_values(rng, n_values)


def test_combined_builtin_trace_proves_and_verifies(stark):
    rng = random.Random(62)
    hashes, signatures, values = small_batch(rng, 4, 2, 40)
    proof = stark.prove_builtins(hashes, signatures, values, n_queries=3, seed=5)
    assert proof["n"] == 2048 and proof["segments"] == ["pedersen", "ecdsa", "rc16"]
    ok, why = S.verify_builtins_proof(proof, hash2=c_hash)
    assert ok, why
    # tampering: a phase-1 cell, the second-phase column, the claimed limb range, a signature, the final layer
    bad = copy.deepcopy(proof)
    bad["queries"][0]["phase1"][0]["values"][15] ^= 1
    assert S.verify_builtins_proof(bad, hash2=c_hash) == (False, "phase 1 path")
    bad = copy.deepcopy(proof)
    bad["queries"][1]["phase2"][2]["value"] ^= 1
    assert S.verify_builtins_proof(bad, hash2=c_hash) == (False, "phase 2 path")
    bad = copy.deepcopy(proof)
    bad["public_inputs"]["rc_max"] -= 1  # a different statement: every challenge changes
    assert not S.verify_builtins_proof(bad, hash2=c_hash)[0]
    bad = copy.deepcopy(proof)
    bad["public_inputs"]["signatures"][0][1] = 0
    assert S.verify_builtins_proof(bad, hash2=c_hash) == (False, "public inputs")
    bad = copy.deepcopy(proof)
    bad["final_layer"][3] ^= 1
    assert not S.verify_builtins_proof(bad, hash2=c_hash)[0]
    bad = copy.deepcopy(proof)
    bad["phase2_root"] ^= 1
    assert not S.verify_builtins_proof(bad, hash2=c_hash)[0]


def test_segment_subsets_and_a_wrong_permutation(stark):
    import torch
    rng = random.Random(63)
    hashes, signatures, values = small_batch(rng, 2, 1, 24)
    for args, segs in (((None, None, values), ["rc16"]), ((hashes, None, values), ["pedersen", "rc16"]),
                       ((None, signatures, None), ["ecdsa"])):
        proof = stark.prove_builtins(*args, n_queries=2, seed=1)
        assert proof["segments"] == segs
        ok, why = S.verify_builtins_proof(proof, hash2=c_hash)
        assert ok, (segs, why)
    # a sorted column that is NOT a permutation of the limb column: the product column cannot close (p_last != 1)
    # and the composition of that trace is not a polynomial of the claimed degree
    padded, lo, hi = stark.rc16_fill(values, 32)
    cols = stark.rc16_columns(stark.felts_to_tensor(padded))
    n = cols.shape[1]
    sorted_limbs = cols[2, :, 0]
    k = int(torch.nonzero(sorted_limbs[1:] != sorted_limbs[:-1])[3]) + 1
    cols[2, k, 0] = cols[2, k - 1, 0]  # one cell of the pool replaced by its predecessor
    z = rng.randrange(P)
    p = stark.rc16_product(cols[0], cols[2], z)
    assert stark.tensor_to_felts(p[-1:])[0] != 1
    alphas = [rng.randrange(P) for _ in range(8)]
    comp = stark.air_eval_rc16(stark.lde(cols), stark.lde(p.unsqueeze(0))[0], stark.periodic_lde(n, 3, "cuda", "rc16"), n,
                               alphas, z, lo, hi)
    assert not S.poly_degree_bound_check(stark.tensor_to_felts(comp), S.GEN, 3 * n - 1)
    # a value of 2^128 is not a range-checked value
    with pytest.raises(AssertionError):
        stark.prove_builtins(None, None, [2**128])
    del torch


def test_builtin_usage_of_a_4096_order_batch_in_one_trace(stark):
    """BASELINE.json configs[2] as a statement: the 4 x 4096 message-hash chain hashes, the 4096 signature
    verifications and 8 range checks per order (the three amounts, nonce, expiration, position id and the two
    fields of the message hash that order/order.cairo:36-56 range-checks) - 16 384 hashes + 4096 verifications
    + 32 768 values in ONE trace of 2^23 rows x 17 columns (+ 1 in the second phase), proved on the GPU,
    verified on the CPU."""
    import workloads as wl
    from starkperp import batch, perpetual_messages as pm
    orders = wl.limit_orders(4096, seed=2)
    keys = wl.private_keys(1024, seed=12)
    words = [pm._limit_order_words(*wl.order_args(o)) for o in orders]
    hash_inputs, acc = [], [w[0] for w in words]
    for k in range(1, 5):
        ys = [w[k] for w in words]
        hash_inputs += list(zip(acc, ys))
        acc = batch.pedersen_hash_many(acc, ys)
    zs = [z % 2**251 for z in acc]
    assert acc == pm.limit_order_msgs_many([wl.order_args(o) for o in orders])
    pubs = batch.public_keys_many(keys)
    sigs = batch.sign_many(zs, [keys[o["key_index"]] for o in orders])
    signatures = [(z, r, s, pubs[o["key_index"]]) for z, (r, s), o in zip(zs, sigs, orders)]
    values = []
    for o, z in zip(orders, acc):
        values += [o["amount_synthetic"], o["amount_collateral"], o["max_amount_fee"], o["nonce"], o["position_id"],
                   o["expiration_timestamp"], z & (2**128 - 1), (z >> 128) & (2**59 - 1)]
    assert len(hash_inputs) == 16384 and len(signatures) == 4096 and len(values) == 32768
    proof = stark.prove_builtins(hash_inputs, signatures, values, n_queries=2, seed=7)
    assert proof["n"] == 1 << 23 and proof["segments"] == ["pedersen", "ecdsa", "rc16"]
    assert proof["public_inputs"]["rc_min"] == 0 and proof["public_inputs"]["rc_max"] == 0xFFFF
    ok, why = S.verify_builtins_proof(proof, hash2=c_hash)
    assert ok, why
    bad = copy.deepcopy(proof)
    bad["queries"][1]["layers"][0][0]["value"] ^= 1
    assert not S.verify_builtins_proof(bad, hash2=c_hash)[0]

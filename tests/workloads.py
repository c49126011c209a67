"""
Seeded synthetic workloads shared by oracle/gen_golden.py (which runs them through the real
reference), the parity tests and bench.py.  Pure `random.Random` so the same inputs appear in the
build container and on the GPU box.  Shapes follow SURVEY.md section 8(d) / BASELINE.json configs.
"""

import hashlib
import random

P = 2**251 + 17 * 2**192 + 1
N = 0x800000000000010FFFFFFFFFFFFFFFFB781126DCAE7B2321E66A241ADC64D2F

EDGE_FELTS = [0, 1, 2**248 - 1, 2**248, 2**251, P - 1]


def digest_felts(values):
    """sha256 of the 32-byte big-endian encodings - how bulk expected outputs are committed."""
    h = hashlib.sha256()
    for v in values:
        h.update(int(v).to_bytes(32, "big"))
    return h.hexdigest()


def pedersen_pairs(n, seed=0):
    rng = random.Random(seed)
    return [(rng.randrange(P), rng.randrange(P)) for _ in range(n)]


def edge_pairs():
    return [(a, b) for a in EDGE_FELTS for b in EDGE_FELTS]


def private_keys(n, seed=10):
    rng = random.Random(seed)
    return [rng.randrange(1, N) for _ in range(n)]


def sign_cases(n, seed=11):
    """(z, d, seed_or_None).  z bit-lengths sweep the RFC-6979 one-nibble pad rule
    (signature.py:119-121)."""
    rng = random.Random(seed)
    out = []
    special_bits = [244, 247, 248, 249, 250, 251, 252 - 1, 8, 1]
    for i in range(n):
        d = rng.randrange(1, N)
        if i < 4 * len(special_bits):
            b = special_bits[i % len(special_bits)]
            z = rng.randrange(1 << (b - 1), 1 << b) if b > 1 else 1
            z = min(z, 2**251 - 1)
        elif i == 4 * len(special_bits):
            z = 0
        else:
            z = rng.randrange(2**251)
        sd = None if i % 4 else rng.randrange(1, 1 << 40)
        out.append((z, d, sd))
    return out


def leaves(n, seed=1):
    rng = random.Random(seed)
    return [rng.randrange(P) for _ in range(n)]


def positions(n, seed=3):
    """(public_key, collateral_balance, [(asset_id, funding_index, balance)...]) with n_assets in
    0..3 and values inside the bounds of definitions/constants.cairo:11-38."""
    rng = random.Random(seed)
    out = []
    for _ in range(n):
        n_assets = rng.randrange(4)
        ids = set()
        while len(ids) < n_assets:
            ids.add(rng.randrange(1, 2**120))
        ids = sorted(ids)
        assets = [
            (a, rng.randrange(-(2**63), 2**63), rng.randrange(-(2**63), 2**63)) for a in ids
        ]
        out.append((rng.randrange(2**251), rng.randrange(-(2**63), 2**63), assets))
    return out


def limit_orders(n, seed=2, n_keys=1024):
    """n limit orders as dicts of the ten perpetual_messages.get_limit_order_msg arguments plus a
    key index; fields uniform in the ranges asserted at perpetual_messages.py:226-236."""
    rng = random.Random(seed)
    collateral = rng.randrange(2**250)
    out = []
    for _ in range(n):
        out.append(
            dict(
                asset_id_synthetic=rng.randrange(2**128),
                asset_id_collateral=collateral,
                is_buying_synthetic=rng.randrange(2),
                asset_id_fee=collateral,
                amount_synthetic=rng.randrange(2**64),
                amount_collateral=rng.randrange(2**64),
                max_amount_fee=rng.randrange(2**64),
                nonce=rng.randrange(2**32),
                position_id=rng.randrange(2**64),
                expiration_timestamp=rng.randrange(2**32),
                key_index=rng.randrange(n_keys),
            )
        )
    return out


ORDER_ARG_NAMES = [
    "asset_id_synthetic", "asset_id_collateral", "is_buying_synthetic", "asset_id_fee",
    "amount_synthetic", "amount_collateral", "max_amount_fee", "nonce", "position_id",
    "expiration_timestamp",
]


def order_args(o):
    return [o[k] for k in ORDER_ARG_NAMES]

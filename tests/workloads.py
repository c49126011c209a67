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


def comb_table_multiples():
    """Integer multiples held by a key's comb table (csrc/ecdsa.hip "Key tables"):
    t_v = 2^224 + sum_{i<7} (+-) 2^(32 i); a sample of the 128 patterns."""
    out = []
    for v in (0, 1, 2, 64, 85, 126, 127):
        out.append(2**224 + sum((1 if (v >> i) & 1 else -1) * 2 ** (32 * i) for i in range(7)))
    return out


def crafted_verify_cases(d, q, rng, extra_u2=()):
    """(z, r, s) triples for the key q = d G whose u1 = z/s and u2 = r/s (mod N) sit on the corners of
    the verification kernels' scalar handling; about a third are VALID signatures (constructed from
    the chosen u1, u2), the rest carry an r that does not match or have u1 G = +-u2 Q."""
    from oracle import ref_py as R
    N = R.EC_ORDER
    u2_targets = [1, 2, 3, 15, 16, 17, 31, 32, 33, N - 1, N - 2, N - 3, 2**251, 2**251 - 1, 2**251 + 1,
                  (N - 1) // 2, (N + 1) // 2, int("f" * 62, 16) % N, int("1" * 62, 16), int("8" * 62, 16) % N]
    u2_targets += [N - 2 * t for t in range(1, 16, 2)] + [2 * t for t in range(1, 16, 2)]
    u2_targets += [1 << k for k in (4, 21, 26, 63, 126, 250)] + [(1 << k) - 1 for k in (21, 26, 252)]
    u2_targets += list(extra_u2)
    u1_targets = [1, 2, 2**21 - 1, 2**21, 2**26, 2**42 - 1, 2**251, N - 1, N - 2, rng.randrange(N)]
    cases = []
    for u2 in u2_targets:
        u2 %= N
        if u2 == 0:
            continue
        # (i) a VALID signature with this u2: pick u1, let R = u1 G + u2 Q, r = x(R), s = r / u2, z = u1 s
        u2q = R.ec_mult(u2, q)
        for u1 in u1_targets + [rng.randrange(1, N) for _ in range(200)]:
            a = R.ec_mult(u1, tuple(R.EC_GEN))
            if a[0] == u2q[0]:
                continue
            r = R.ec_add(a, u2q)[0]
            if not 1 <= r < 2**251:
                continue
            s = r * pow(u2, -1, N) % N
            z = u1 * s % N
            if s and z < 2**251 and 1 <= pow(s, -1, N) < 2**251:
                cases.append((z, r, s))
                break
        # (ii) the same scalars with an r that does not match, and u1 G = +-u2 Q
        for u1 in (rng.randrange(1, N), u2 * d % N, (N - u2 * d) % N):
            for _ in range(200):
                s = rng.randrange(1, N)
                r, z = u2 * s % N, u1 * s % N
                if 1 <= r < 2**251 and z < 2**251 and 1 <= pow(s, -1, N) < 2**251:
                    cases.append((z, r, s))
                    break
    return cases


def extreme_felts(prime=None):
    """Field elements whose 29-bit limb strings are what random data never looks like: every limb at its maximum,
    alternating empty / full limbs, the neighbours of 0, p, 2^251 and of the limb boundaries - the patterns on
    which a lazy (carry-only) reduction overflows first (the B = 4 bug of the NTT)."""
    P = prime or (2**251 + 17 * 2**192 + 1)
    full = (1 << 232) - 1                                  # limbs 0..7 all 2^29 - 1
    vals = [0, 1, 2, P - 1, P - 2, (P - 1) // 2, (P + 1) // 2, 1 << 251, (1 << 251) - 1, (1 << 251) + 1,
            full, full + (((1 << 19) - 1) << 232), full + (1 << 250), 17 << 192, (17 << 192) - 1, (17 << 192) + 1]
    alt = sum(((1 << 29) - 1) << (29 * k) for k in range(0, 8, 2))
    vals += [alt, alt << 29, alt + (((1 << 19) - 1) << 232), (alt << 29) + (1 << 250)]
    for k in (28, 29, 30, 57, 58, 59, 87, 116, 174, 191, 192, 193, 203, 231, 232, 233, 250):
        vals += [(1 << k) - 1, 1 << k, P - (1 << k), P - (1 << k) - 1]
    return sorted(set(v % P for v in vals))

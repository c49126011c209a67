"""
ORACLE (test infrastructure, NOT product code).

Dependency-free pure-Python restatement of the reference's Pedersen / Stark-ECDSA arithmetic and
of the message / leaf / tree conventions that sit either side of it.  Only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the product path
(stark-perpetual_amd/) never does.

Parity status: PINNED.  tests/test_oracle_golden.py checks this file against
  * the reference's own vectors (hash_test, keys_precomputed, party_a_order signature, the four
    perpetual_messages KATs) copied as data into tests/golden/reference_kats.json, and
  * vectors generated in the build container by importing the reference itself
    (oracle/gen_golden.py -> tests/golden/*.json).
Tree conventions (merkle_root / merkle_multi_update) restate public cairo-lang behaviour whose
implementation is absent from the reference tree: those two functions are "parity unpinned"
beyond the fact that every node is a pinned pedersen_hash.

Each function cites the reference file:line (relative to /root/reference/src) it follows.
"""

import hashlib
import hmac

# --------------------------------------------------------------------------------------------
# Parameters: starkware/crypto/signature/pedersen_params.json:20-25, signature.py:41-68
# --------------------------------------------------------------------------------------------
FIELD_PRIME = 2**251 + 17 * 2**192 + 1
FIELD_GEN = 3
ALPHA = 1
BETA = 0x6F21413EFBE40DE150E596D72F7A8C5609AD26C15C915C1F4CDFCB99CEE9E89
EC_ORDER = 0x800000000000010FFFFFFFFFFFFFFFFB781126DCAE7B2321E66A241ADC64D2F
N_ELEMENT_BITS_ECDSA = 251
N_ELEMENT_BITS_HASH = 252

# The six independent points of the table; the other 500 entries are doublings of P0..P3
# (nothing_up_my_sleeve_gen.py:88-90).  Pinned by the sha256 of the expanded table in
# tests/golden/params_digest.json.
_SHIFT = (
    0x49EE3EBA8C1600700EE1B87EB599F16716B0B1022947733551FDE4050CA6804,
    0x3CA0CFE4B3BC6DDF346D49D06EA0ED34E621062C0E056C1D0405D266E10268A,
)
_GEN = (
    0x1EF15C18599971B7BECED415A40F0C7DEACFD9B0D1819E03D723D8BC943CFCA,
    0x5668060AA49730B7BE4801DF46EC62DE53ECD11ABE43A32873000C36E8DC1F,
)
_P0 = (
    0x234287DCBAFFE7F969C748655FCA9E58FA8120B6D56EB0C1080D17957EBE47B,
    0x3B056F100F96FB21E889527D41F4E39940135DD7A6C94CC6ED0268EE89E5615,
)
_P1 = (
    0x4FA56F376C83DB33F9DAB2656558F3399099EC1DE5E3018B7A6932DBA8AA378,
    0x3FA0984C931C9E38113E0C0E47E4401562761F92A7A23B45168F4E80FF5B54D,
)
_P2 = (
    0x4BA4CC166BE8DEC764910F75B45F74B40C690C74709E90F3AA372F0BD2D6997,
    0x40301CF5C1751F4B971E46C4EDE85FCAC5C59A5CE5AE7C48151F27B24B219C,
)
_P3 = (
    0x54302DCB0E6CC1C6E44CCA8F61A63BB2CA65048D53FB325D36FF12C49A58202,
    0x1B77B3E37D13504B348046268D8AE25CE98AD783C25561A879DCC77E99C2426,
)


# --------------------------------------------------------------------------------------------
# Field / curve arithmetic: starkware/crypto/signature/math_utils.py:50-100
# --------------------------------------------------------------------------------------------
def _egcd_inverse(a, m):
    """Inverse of a mod m by the extended Euclidean algorithm (what sympy's igcdex does for
    math_utils.py:54).  Raises AssertionError when gcd != 1 (math_utils.py:55)."""
    a %= m
    r0, r1 = m, a
    t0, t1 = 0, 1
    while r1:
        q = r0 // r1
        r0, r1 = r1, r0 - q * r1
        t0, t1 = t1, t0 - q * t1
    assert r0 == 1
    return t0 % m


def div_mod(n, m, p):
    """math_utils.py:50-56."""
    return (n * _egcd_inverse(m, p)) % p


def ec_add(pt1, pt2, p=FIELD_PRIME):
    """math_utils.py:59-68 (affine chord; x's must differ)."""
    x1, y1 = pt1
    x2, y2 = pt2
    assert (x1 - x2) % p != 0
    lam = div_mod(y1 - y2, x1 - x2, p)
    x3 = (lam * lam - x1 - x2) % p
    return x3, (lam * (x1 - x3) - y1) % p


def ec_neg(pt, p=FIELD_PRIME):
    """math_utils.py:71-76."""
    return pt[0], (-pt[1]) % p


def ec_double(pt, alpha=ALPHA, p=FIELD_PRIME):
    """math_utils.py:79-88 (affine tangent; y must be non-zero)."""
    x, y = pt
    assert y % p != 0
    lam = div_mod(3 * x * x + alpha, 2 * y, p)
    x3 = (lam * lam - 2 * x) % p
    return x3, (lam * (x - x3) - y) % p


def ec_mult(m, pt, alpha=ALPHA, p=FIELD_PRIME):
    """math_utils.py:91-100.  The reference recurses; this walks the same add/double sequence
    iteratively (the recursion unwinds into: process the bits of m from the top, doubling the
    *base* on the way down).  Same group element, same intermediate asserts."""
    # Recursion shape: m even -> mult(m/2, 2P); m odd -> mult(m-1, P) + P; m == 1 -> P.
    pending = []  # points to add after the inner call returns
    while m != 1:
        if m % 2 == 0:
            m //= 2
            pt = ec_double(pt, alpha, p)
        else:
            m -= 1
            pending.append(pt)
    acc = pt
    for q in reversed(pending):
        acc = ec_add(acc, q, p)
    return acc


def _expand_constant_points():
    pts = [_SHIFT, _GEN]
    for base, n in ((_P0, 248), (_P1, 4), (_P2, 248), (_P3, 4)):
        q = base
        for _ in range(n):
            pts.append(q)
            q = ec_double(q)
    return [list(q) for q in pts]


CONSTANT_POINTS = _expand_constant_points()
assert len(CONSTANT_POINTS) == 506
SHIFT_POINT = CONSTANT_POINTS[0]
MINUS_SHIFT_POINT = (SHIFT_POINT[0], FIELD_PRIME - SHIFT_POINT[1])
EC_GEN = CONSTANT_POINTS[1]


def constant_points_digest():
    """sha256 over the 506 points, each coordinate as 32 big-endian bytes."""
    h = hashlib.sha256()
    for x, y in CONSTANT_POINTS:
        h.update(x.to_bytes(32, "big") + y.to_bytes(32, "big"))
    return h.hexdigest()


# --------------------------------------------------------------------------------------------
# Pedersen hash: signature.py:296-318
# --------------------------------------------------------------------------------------------
def pedersen_hash_as_point(*elements):
    """signature.py:300-318: LSB-first 252-step conditional affine add per element."""
    acc = tuple(SHIFT_POINT)
    for i, x in enumerate(elements):
        assert 0 <= x < FIELD_PRIME
        seg = CONSTANT_POINTS[2 + 252 * i : 2 + 252 * (i + 1)]
        assert len(seg) == N_ELEMENT_BITS_HASH
        for q in seg:
            assert acc[0] != q[0], "Unhashable input."
            if x & 1:
                acc = ec_add(acc, q)
            x >>= 1
        assert x == 0
    return acc


def pedersen_hash(*elements):
    """signature.py:296-297."""
    return pedersen_hash_as_point(*elements)[0]


# --------------------------------------------------------------------------------------------
# Square roots: math_utils.py:36-47 (sympy.is_quad_residue / sqrt_mod(all_roots) -> min)
# --------------------------------------------------------------------------------------------
def is_quad_residue(n, p=FIELD_PRIME):
    n %= p
    return n == 0 or pow(n, (p - 1) // 2, p) == 1


def sqrt_mod(n, p=FIELD_PRIME):
    """Smallest non-negative root of m*m = n (mod p); Tonelli-Shanks (p - 1 = 2^192 * odd)."""
    n %= p
    if n == 0:
        return 0
    assert is_quad_residue(n, p)
    q, s = p - 1, 0
    while q % 2 == 0:
        q //= 2
        s += 1
    z = 2
    while is_quad_residue(z, p):
        z += 1
    c = pow(z, q, p)
    r = pow(n, (q + 1) // 2, p)
    t = pow(n, q, p)
    m = s
    while t != 1:
        i, t2 = 0, t
        while t2 != 1:
            t2 = t2 * t2 % p
            i += 1
        b = pow(c, 1 << (m - i - 1), p)
        r = r * b % p
        c = b * b % p
        t = t * c % p
        m = i
    return min(r, p - r)


# --------------------------------------------------------------------------------------------
# Where the constants come from: nothing_up_my_sleeve_gen.py:50-91 (digits of pi)
# --------------------------------------------------------------------------------------------
def pi_digits(n_digits):
    """The first n_digits decimal digits of pi as a string ("3141...", truncated, no rounding);
    stands in for math_utils.pi_as_string (:28-33, mpmath), whose caller asks for 100 spare digits
    and reads only the leading ones.  Machin: pi = 16 atan(1/5) - 4 atan(1/239), integers only."""
    guard = 10**15
    scale = 10 ** (n_digits - 1) * guard

    def atan_inv(q):
        total = term = scale // q
        k, sign = 3, -1
        while term:
            term //= q * q
            total += sign * (term // k)
            k, sign = k + 2, -sign
        return total

    return str((16 * atan_inv(5) - 4 * atan_inv(239)) // guard)


def generate_constant_points(n_points=6):
    """nothing_up_my_sleeve_gen.py:50-91: beta = first 76 digits of pi + 379; base point i takes
    its x from the i-th block of 76 digits (incremented until on the curve, y = the smaller root);
    points 1, 2 (shift, generator) are kept as they are, point i > 2 is expanded into its 248
    (i odd) or 4 (i even) successive doublings.  Returns (beta, table)."""
    digits = pi_digits(76 * (1 + n_points) + 100)
    beta = int(digits[:76]) + 379
    table, i = [], 0
    while i < n_points:
        i += 1
        x = int(digits[76 * i : 76 * (i + 1)])
        while not is_quad_residue(x**3 + ALPHA * x + beta, FIELD_PRIME):
            x += 1
        pt = (x % FIELD_PRIME, sqrt_mod(x**3 + ALPHA * x + beta, FIELD_PRIME))
        if i <= 2:
            table.append(pt)
            continue
        for _ in range(248 if i % 2 == 1 else 4):
            table.append(pt)
            pt = ec_double(pt, ALPHA, FIELD_PRIME)
    return beta, table


# --------------------------------------------------------------------------------------------
# ECDSA: signature.py:79-260
# --------------------------------------------------------------------------------------------
class InvalidPublicKeyError(Exception):
    """signature.py:79-81."""

    def __init__(self):
        super().__init__("Given x coordinate does not represent any point on the elliptic curve.")


def get_y_coordinate(stark_key_x_coordinate):
    """signature.py:84-96."""
    x = stark_key_x_coordinate
    rhs = (x * x * x + ALPHA * x + BETA) % FIELD_PRIME
    if not is_quad_residue(rhs):
        raise InvalidPublicKeyError()
    return sqrt_mod(rhs)


def private_key_to_ec_point_on_stark_curve(priv_key):
    """signature.py:104-106."""
    assert 0 < priv_key < EC_ORDER
    return ec_mult(priv_key, tuple(EC_GEN))


def private_to_stark_key(priv_key):
    """signature.py:109-110."""
    return private_key_to_ec_point_on_stark_curve(priv_key)[0]


def inv_mod_curve_size(x):
    """signature.py:113-114."""
    return div_mod(1, x, EC_ORDER)


def _rfc6979_k(order, secexp, data, extra_entropy=b""):
    """RFC 6979 section 3.2 with HMAC-SHA256, python-ecdsa 0.17 conventions (the third-party
    dependency behind signature.py:25,128-134; pinned here by the party_a_order signature KAT and
    by RFC 6979 A.2.5 in tests/test_oracle_golden.py)."""
    qlen = order.bit_length()
    rolen = (qlen + 7) // 8
    olen = (len("%x" % order) + 1) // 2

    def bits2int(b):
        v = int.from_bytes(b, "big")
        extra = len(b) * 8 - qlen
        return v >> extra if extra > 0 else v

    z = bits2int(data)
    if z >= order:
        z -= order
    seed = secexp.to_bytes(olen, "big") + z.to_bytes(olen, "big") + extra_entropy
    mac = lambda key, msg: hmac.new(key, msg, hashlib.sha256).digest()
    v = b"\x01" * 32
    k = b"\x00" * 32
    k = mac(k, v + b"\x00" + seed)
    v = mac(k, v)
    k = mac(k, v + b"\x01" + seed)
    v = mac(k, v)
    while True:
        t = b""
        while len(t) < rolen:
            v = mac(k, v)
            t += v
        cand = bits2int(t)
        if 1 <= cand < order:
            return cand
        k = mac(k, v + b"\x00")
        v = mac(k, v)


def generate_k_rfc6979(msg_hash, priv_key, seed=None):
    """signature.py:117-134 (one-nibble pad rule :119-121; seed -> extra_entropy :123-126)."""
    bl = msg_hash.bit_length()
    if 1 <= bl % 8 <= 4 and bl >= 248:
        msg_hash *= 16
    extra = b"" if seed is None else seed.to_bytes((seed.bit_length() + 7) // 8, "big")
    data = msg_hash.to_bytes((msg_hash.bit_length() + 7) // 8, "big")
    return _rfc6979_k(EC_ORDER, priv_key, data, extra)


def sign(msg_hash, priv_key, seed=None):
    """signature.py:137-173."""
    assert 0 <= msg_hash < 2**N_ELEMENT_BITS_ECDSA, "Message not signable."
    while True:
        k = generate_k_rfc6979(msg_hash, priv_key, seed)
        seed = 1 if seed is None else seed + 1
        r = ec_mult(k, tuple(EC_GEN))[0]
        if not (1 <= r < 2**N_ELEMENT_BITS_ECDSA):
            continue
        if (msg_hash + r * priv_key) % EC_ORDER == 0:
            continue
        w = div_mod(k, msg_hash + r * priv_key, EC_ORDER)
        if not (1 <= w < 2**N_ELEMENT_BITS_ECDSA):
            continue
        return r, inv_mod_curve_size(w)


def mimic_ec_mult_air(m, point, shift_point):
    """signature.py:176-190."""
    assert 0 < m < 2**N_ELEMENT_BITS_ECDSA
    acc = tuple(shift_point)
    point = tuple(point)
    for _ in range(N_ELEMENT_BITS_ECDSA):
        assert acc[0] != point[0]
        if m & 1:
            acc = ec_add(acc, point)
        point = ec_double(point)
        m >>= 1
    assert m == 0
    return acc


def is_point_on_curve(x, y):
    """signature.py:193-194."""
    return pow(y, 2, FIELD_PRIME) == (pow(x, 3, FIELD_PRIME) + ALPHA * x + BETA) % FIELD_PRIME


def is_valid_stark_private_key(private_key):
    """signature.py:197-201."""
    return 0 < private_key < EC_ORDER


def is_valid_stark_key(stark_key):
    """signature.py:204-214."""
    try:
        get_y_coordinate(stark_key)
    except InvalidPublicKeyError:
        return False
    return True


def verify(msg_hash, r, s, public_key):
    """signature.py:217-260."""
    assert 1 <= s < EC_ORDER, "s = %s" % s
    w = inv_mod_curve_size(s)
    assert 1 <= r < 2**N_ELEMENT_BITS_ECDSA, "r = %s" % r
    assert 1 <= w < 2**N_ELEMENT_BITS_ECDSA, "w = %s" % w
    assert 0 <= msg_hash < 2**N_ELEMENT_BITS_ECDSA, "msg_hash = %s" % msg_hash
    if isinstance(public_key, int):
        try:
            y = get_y_coordinate(public_key)
        except InvalidPublicKeyError:
            return False
        return verify(msg_hash, r, s, (public_key, y)) or verify(
            msg_hash, r, s, (public_key, (-y) % FIELD_PRIME)
        )
    assert is_point_on_curve(public_key[0], public_key[1])
    try:
        zg = mimic_ec_mult_air(msg_hash, EC_GEN, MINUS_SHIFT_POINT)
        rq = mimic_ec_mult_air(r, public_key, SHIFT_POINT)
        wb = mimic_ec_mult_air(w, ec_add(zg, rq), SHIFT_POINT)
        x = ec_add(wb, MINUS_SHIFT_POINT)[0]
    except AssertionError:
        return False
    return r == x


def grind_key(key_seed, key_value_limit):
    """signature.py:263-288."""
    ceiling = 2**256 - (2**256 % key_value_limit)

    def enc(v):
        return v.to_bytes(max(1, (v.bit_length() + 7) // 8), "big")

    index = 0
    while True:
        key = int.from_bytes(hashlib.sha256(enc(key_seed) + enc(index)).digest(), "big")
        if key < ceiling:
            return key % key_value_limit
        index += 1


# --------------------------------------------------------------------------------------------
# Signed-message packers: services/perpetual/public/perpetual_messages.py
# --------------------------------------------------------------------------------------------
LIMIT_ORDER_WITH_FEES = 3
TRANSFER = 4
CONDITIONAL_TRANSFER = 5
WITHDRAWAL = 6
WITHDRAWAL_TO_ADDRESS = 7


def _pack(fields):
    """fields = [(value, width_bits), ...] most-significant first."""
    acc = 0
    for value, width in fields:
        acc = (acc << width) + value
    return acc


def limit_order_words(
    asset_id_synthetic, asset_id_collateral, is_buying_synthetic, asset_id_fee,
    amount_synthetic, amount_collateral, max_amount_fee, nonce, position_id,
    expiration_timestamp,
):
    """The four hash inputs (a, b, c, w0, w1) of perpetual_messages.py:253-286:
    msg = H(H(H(H(a,b),c),w0),w1)."""
    if is_buying_synthetic:
        sell, buy = asset_id_collateral, asset_id_synthetic
        amt_sell, amt_buy = amount_collateral, amount_synthetic
    else:
        sell, buy = asset_id_synthetic, asset_id_collateral
        amt_sell, amt_buy = amount_synthetic, amount_collateral
    w0 = _pack([(amt_sell, 0), (amt_buy, 64), (max_amount_fee, 64), (nonce, 32)])
    w1 = _pack(
        [(LIMIT_ORDER_WITH_FEES, 0), (position_id, 64), (position_id, 64), (position_id, 64),
         (expiration_timestamp, 32), (0, 17)]
    )
    return sell, buy, asset_id_fee, w0, w1


def get_limit_order_msg(*args, hash_function=pedersen_hash):
    """perpetual_messages.py:212-286 (bounds :226-236)."""
    (syn, col, _buy, fee, a_syn, a_col, a_fee, nonce, pos, exp) = args
    assert 0 <= syn < 2**128 and 0 <= col < 2**250 and 0 <= fee < 2**250
    assert 0 <= a_syn < 2**64 and 0 <= a_col < 2**64 and 0 <= a_fee < 2**64
    assert 0 <= nonce < 2**32 and 0 <= pos < 2**64 and 0 <= exp < 2**32
    a, b, c, w0, w1 = limit_order_words(*args)
    h = hash_function
    return h(h(h(h(a, b), c), w0), w1)


def get_transfer_msg(
    asset_id, asset_id_fee, receiver_public_key, sender_position_id, receiver_position_id,
    src_fee_position_id, nonce, amount, max_amount_fee, expiration_timestamp,
    hash_function=pedersen_hash,
):
    """perpetual_messages.py:97-162."""
    h = hash_function
    w0 = _pack([(sender_position_id, 0), (receiver_position_id, 64), (src_fee_position_id, 64),
                (nonce, 32)])
    w1 = _pack([(TRANSFER, 0), (amount, 64), (max_amount_fee, 64), (expiration_timestamp, 32),
                (0, 81)])
    return h(h(h(h(asset_id, asset_id_fee), receiver_public_key), w0), w1)


def get_conditional_transfer_msg(
    asset_id, asset_id_fee, receiver_public_key, condition, sender_position_id,
    receiver_position_id, src_fee_position_id, nonce, amount, max_amount_fee,
    expiration_timestamp, hash_function=pedersen_hash,
):
    """perpetual_messages.py:24-94."""
    h = hash_function
    w0 = _pack([(sender_position_id, 0), (receiver_position_id, 64), (src_fee_position_id, 64),
                (nonce, 32)])
    w1 = _pack([(CONDITIONAL_TRANSFER, 0), (amount, 64), (max_amount_fee, 64),
                (expiration_timestamp, 32), (0, 81)])
    return h(h(h(h(h(asset_id, asset_id_fee), receiver_public_key), condition), w0), w1)


def get_withdrawal_to_address_msg(
    asset_id_collateral, position_id, eth_address, nonce, expiration_timestamp, amount,
    hash_function=pedersen_hash,
):
    """perpetual_messages.py:165-209."""
    h = hash_function
    w = _pack([(WITHDRAWAL_TO_ADDRESS, 0), (position_id, 64), (nonce, 32), (amount, 64),
               (expiration_timestamp, 32), (0, 49)])
    return h(h(asset_id_collateral, int(eth_address, 16)), w)


def get_withdrawal_msg(
    asset_id_collateral, position_id, nonce, expiration_timestamp, amount, hash_function=pedersen_hash,
):
    """Old-API withdrawal, transaction type 6: services/perpetual/cairo/transactions/withdrawal.cairo:57-60
    (owner_key == public_key branch) with the packing of :66-74; JS twin
    services/perpetual/public/js/perpetual_messages.js:49-82 (bounds :63-72)."""
    assert 0 <= asset_id_collateral < 2**250 and 0 <= nonce < 2**32 and 0 <= position_id < 2**64
    assert 0 <= expiration_timestamp < 2**32 and 0 <= amount < 2**64
    w = _pack([(WITHDRAWAL, 0), (position_id, 64), (nonce, 32), (amount, 64),
               (expiration_timestamp, 32), (0, 49)])
    return hash_function(asset_id_collateral, w)


def withdrawal_hash(
    asset_id_collateral, position_id, owner_key, public_key, nonce, expiration_timestamp, amount,
    hash_function=pedersen_hash,
):
    """withdrawal.cairo:47-78: both branches of the message the program verifies."""
    h = hash_function
    if owner_key == public_key:
        first, kind = asset_id_collateral, WITHDRAWAL
    else:
        first, kind = h(asset_id_collateral, owner_key), WITHDRAWAL_TO_ADDRESS
    w = kind
    for value, upper in ((position_id, 2**64), (nonce, 2**32), (amount, 2**64),
                         (expiration_timestamp, 2**32)):
        w = w * upper + value
    return h(first, w * 2**49)


def get_price_msg(oracle_name, asset_pair, timestamp, price, hash_function=pedersen_hash):
    """perpetual_messages.py:311-326."""
    assert 0 <= oracle_name < 2**40 and 0 <= asset_pair < 2**128
    assert 0 <= timestamp < 2**32 and 0 <= price < 2**120
    return hash_function((asset_pair << 40) + oracle_name, (price << 32) + timestamp)


# --------------------------------------------------------------------------------------------
# Multi-asset order: services/exchange/cairo/signature_message_hashes.cairo:171-471
# (the Cairo program is the only statement of this format; no Python twin exists)
# --------------------------------------------------------------------------------------------
MULTI_ASSET_OFFCHAIN_ORDER_TYPE = 6


def multi_asset_order_words(signer_key, nonce, expiration_timestamp, system_id, give, receive, conditions):
    """The felts of the hash chain, in chain order (signature_message_hashes.cairo:405-468).
    give / receive: lists of (vault_id, public_key, asset_id, amount) in the field order of
    `VaultInfo` (:172-178).  Linearisation (:290-329, :405-431): receive first, then give; per entry
    asset -> `assets`, (vault, amount) -> `vaults_and_amounts`, and when the entry's key differs from
    the signer's its key and its index IN ITS OWN LIST -> `third_party_*`."""
    vaults_and_amounts, assets, third_keys, third_idx = [], [], [], []
    for entries in (receive, give):
        for index, (vault_id, public_key, asset_id, amount) in enumerate(entries):
            assets.append(asset_id)
            vaults_and_amounts += [vault_id, amount]
            if public_key != signer_key:
                third_idx.append(index)
                third_keys.append(public_key)
    words = list(conditions) + assets + third_keys
    for i in range(0, len(vaults_and_amounts), 3):  # three 64-bit fields per felt (:264-288)
        acc = 0
        for v in vaults_and_amounts[i : i + 3]:
            acc = acc * 2**64 + v
        words.append(acc)
    for i in range(0, len(third_idx), 20):  # twenty 12-bit indices per felt (:205-260)
        acc = 0
        for v in third_idx[i : i + 20]:
            acc = acc * 2**12 + v
        words.append(acc)
    meta = MULTI_ASSET_OFFCHAIN_ORDER_TYPE  # :433-463
    meta = meta * 2**32 + nonce
    meta = meta * 2**32 + expiration_timestamp
    meta = meta * 2**12 + len(give)
    meta = meta * 2**12 + len(receive)
    meta = meta * 2**12 + len(third_idx)
    meta = meta * 2**12 + len(conditions)
    meta = meta * 2**126 + system_id
    words.append(meta * 2**3)
    return words


def multi_asset_order_hash(signer_key, nonce, expiration_timestamp, system_id, give, receive, conditions,
                           hash_function=pedersen_hash):
    """signature_message_hashes.cairo:387-471: hash_felts_no_padding(words[1:], initial_hash=words[0]),
    i.e. the left fold h(...h(h(w0, w1), w2)..., packed_metadata) (:465-470; hash_felts_no_padding is
    cairo-lang's plain hash2 fold over the data with the given initial value)."""
    words = multi_asset_order_words(signer_key, nonce, expiration_timestamp, system_id, give, receive,
                                    conditions)
    acc = words[0]
    for w in words[1:]:
        acc = hash_function(acc, w)
    return acc


# --------------------------------------------------------------------------------------------
# Position leaf: services/perpetual/cairo/position/hash.cairo:22-74,
# bounds services/perpetual/cairo/definitions/constants.cairo:11-38
# --------------------------------------------------------------------------------------------
BALANCE_LOWER_BOUND = -(2**63)
BALANCE_UPPER_BOUND = 2**63
FUNDING_INDEX_LOWER_BOUND = -(2**63)
FUNDING_INDEX_UPPER_BOUND = 2**63
N_ASSETS_UPPER_BOUND = 2**16


def position_hash(public_key, collateral_balance, assets, hash_function=pedersen_hash):
    """hash.cairo:58-74; assets = [(asset_id, cached_funding_index, balance), ...] sorted by id."""
    h = hash_function
    acc = 0
    for asset_id, funding, balance in assets:
        packed = asset_id
        packed = packed * (FUNDING_INDEX_UPPER_BOUND - FUNDING_INDEX_LOWER_BOUND) + (
            funding - FUNDING_INDEX_LOWER_BOUND)
        packed = packed * (BALANCE_UPPER_BOUND - BALANCE_LOWER_BOUND) + (
            balance - BALANCE_LOWER_BOUND)
        acc = h(acc, packed)
    acc = h(acc, public_key)
    tail = (collateral_balance - BALANCE_LOWER_BOUND) * N_ASSETS_UPPER_BOUND + len(assets)
    return h(acc, tail)


# --------------------------------------------------------------------------------------------
# Trees (cairo-lang merkle_multi_update convention; call sites state/state.cairo:155-173;
# hint-side helper starkware/python/merkle_tree.py:4-26).  Node = pedersen_hash(left, right).
# --------------------------------------------------------------------------------------------
def merkle_levels(leaves, hash_function=pedersen_hash):
    """All levels bottom-up of the full binary tree over len(leaves) = 2^h leaves."""
    n = len(leaves)
    assert n >= 1 and n & (n - 1) == 0
    levels = [list(leaves)]
    while len(levels[-1]) > 1:
        cur = levels[-1]
        levels.append([hash_function(cur[2 * i], cur[2 * i + 1]) for i in range(len(cur) // 2)])
    return levels


def merkle_root(leaves, hash_function=pedersen_hash):
    return merkle_levels(leaves, hash_function)[-1][0]


def empty_subtree_roots(height, empty_leaf=0, hash_function=pedersen_hash):
    """roots[k] = root of an all-`empty_leaf` subtree of height k, k = 0..height."""
    roots = [empty_leaf]
    for _ in range(height):
        roots.append(hash_function(roots[-1], roots[-1]))
    return roots


def merkle_multi_update_sparse(height, modifications, empty_leaf=0, hash_function=pedersen_hash):
    """Root after writing {index: leaf} into an otherwise all-empty tree of the given height:
    the induced-subtree walk of merkle_tree.py:18-26 (parents = set(index // 2)), siblings
    outside the induced subtree are empty-subtree roots."""
    empties = empty_subtree_roots(height, empty_leaf, hash_function)
    layer = dict(modifications)
    if not layer:
        return empties[height]
    for level in range(height):
        parents = sorted(set(i // 2 for i in layer))
        layer = {
            i: hash_function(layer.get(2 * i, empties[level]), layer.get(2 * i + 1, empties[level]))
            for i in parents
        }
    assert list(layer) == [0]
    return layer[0]

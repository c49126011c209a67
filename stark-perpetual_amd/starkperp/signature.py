"""Drop-in mirror of the reference module starkware/crypto/signature/signature.py: same names,
signatures, return types and error behaviour; the arithmetic of pedersen_hash / sign / verify /
key derivation runs on the MI355X through libstarkperp (ctypes, include/starkperp.h).

Line references below are to /root/reference/src/starkware/crypto/signature/signature.py.
"""
import hashlib
import itertools
import math
import secrets
from typing import Optional, Tuple, Union

from . import batch
from .math_utils import ECPoint, div_mod, ec_add, ec_double, is_quad_residue, sqrt_mod
from .rfc6979 import generate_k

# ---- parameters (:38-68; values of pedersen_params.json:20-25) ---------------------------------
FIELD_PRIME = 2**251 + 17 * 2**192 + 1
FIELD_GEN = 3
ALPHA = 1
BETA = 0x6F21413EFBE40DE150E596D72F7A8C5609AD26C15C915C1F4CDFCB99CEE9E89
EC_ORDER = 0x800000000000010FFFFFFFFFFFFFFFFB781126DCAE7B2321E66A241ADC64D2F


def _expand_constant_points():
    """The 506-entry table: shift point, generator, then 248 + 4 doublings of each of the two
    per-element base-point pairs (nothing_up_my_sleeve_gen.py:88-90)."""
    seeds = [
        (0x49EE3EBA8C1600700EE1B87EB599F16716B0B1022947733551FDE4050CA6804,
         0x3CA0CFE4B3BC6DDF346D49D06EA0ED34E621062C0E056C1D0405D266E10268A),
        (0x1EF15C18599971B7BECED415A40F0C7DEACFD9B0D1819E03D723D8BC943CFCA,
         0x5668060AA49730B7BE4801DF46EC62DE53ECD11ABE43A32873000C36E8DC1F),
    ]
    runs = [
        ((0x234287DCBAFFE7F969C748655FCA9E58FA8120B6D56EB0C1080D17957EBE47B,
          0x3B056F100F96FB21E889527D41F4E39940135DD7A6C94CC6ED0268EE89E5615), 248),
        ((0x4FA56F376C83DB33F9DAB2656558F3399099EC1DE5E3018B7A6932DBA8AA378,
          0x3FA0984C931C9E38113E0C0E47E4401562761F92A7A23B45168F4E80FF5B54D), 4),
        ((0x4BA4CC166BE8DEC764910F75B45F74B40C690C74709E90F3AA372F0BD2D6997,
          0x40301CF5C1751F4B971E46C4EDE85FCAC5C59A5CE5AE7C48151F27B24B219C), 248),
        ((0x54302DCB0E6CC1C6E44CCA8F61A63BB2CA65048D53FB325D36FF12C49A58202,
          0x1B77B3E37D13504B348046268D8AE25CE98AD783C25561A879DCC77E99C2426), 4),
    ]
    table = [list(pt) for pt in seeds]
    for start, count in runs:
        pt = start
        for _ in range(count):
            table.append(list(pt))
            pt = ec_double(pt, ALPHA, FIELD_PRIME)
    return table


CONSTANT_POINTS = _expand_constant_points()
PEDERSEN_PARAMS = {
    "FIELD_PRIME": FIELD_PRIME, "FIELD_GEN": FIELD_GEN, "ALPHA": ALPHA, "BETA": BETA,
    "EC_ORDER": EC_ORDER, "CONSTANT_POINTS": CONSTANT_POINTS,
}

N_ELEMENT_BITS_ECDSA = math.floor(math.log(FIELD_PRIME, 2))
assert N_ELEMENT_BITS_ECDSA == 251
N_ELEMENT_BITS_HASH = FIELD_PRIME.bit_length()
assert N_ELEMENT_BITS_HASH == 252
assert 2**N_ELEMENT_BITS_ECDSA < EC_ORDER < FIELD_PRIME

SHIFT_POINT = CONSTANT_POINTS[0]
MINUS_SHIFT_POINT = (SHIFT_POINT[0], FIELD_PRIME - SHIFT_POINT[1])
EC_GEN = CONSTANT_POINTS[1]

ECSignature = Tuple[int, int]


class InvalidPublicKeyError(Exception):
    """:79-81."""

    def __init__(self):
        super().__init__("Given x coordinate does not represent any point on the elliptic curve.")


def get_y_coordinate(stark_key_x_coordinate: int) -> int:
    """:84-96.  Host-side helper (big-int square root); verify() does not need it - the GPU kernel
    tests both y candidates without extracting a root."""
    x = stark_key_x_coordinate
    y_squared = (x * x * x + ALPHA * x + BETA) % FIELD_PRIME
    if not is_quad_residue(y_squared, FIELD_PRIME):
        raise InvalidPublicKeyError()
    return sqrt_mod(y_squared, FIELD_PRIME)


def get_random_private_key() -> int:
    """:99-101."""
    return secrets.randbelow(EC_ORDER - 1) + 1


def private_key_to_ec_point_on_stark_curve(priv_key: int) -> ECPoint:
    """:104-106 - d * EC_GEN on the GPU (sp_public_key_batch)."""
    assert 0 < priv_key < EC_ORDER
    return batch.public_keys_many([priv_key])[0]


def private_to_stark_key(priv_key: int) -> int:
    """:109-110."""
    return private_key_to_ec_point_on_stark_curve(priv_key)[0]


def inv_mod_curve_size(x: int) -> int:
    """:113-114."""
    return div_mod(1, x, EC_ORDER)


def generate_k_rfc6979(msg_hash: int, priv_key: int, seed: Optional[int] = None) -> int:
    """:117-134.  RFC 6979 nonce for (message, key); `seed` becomes the extra entropy of the retry."""
    bits = msg_hash.bit_length()
    if bits >= 248 and 1 <= bits % 8 <= 4:
        msg_hash <<= 4  # one-nibble pad, elliptic.js convention (:119-121)
    entropy = b"" if seed is None else seed.to_bytes((seed.bit_length() + 7) // 8, "big")
    message = msg_hash.to_bytes((msg_hash.bit_length() + 7) // 8, "big")
    return generate_k(EC_ORDER, priv_key, hashlib.sha256, message, extra_entropy=entropy)


def sign(msg_hash: int, priv_key: int, seed: Optional[int] = None) -> ECSignature:
    """:137-173.  RFC 6979 nonce, k * EC_GEN, the mod-N finish and the retry of a rejected nonce
    (:158-170) with the next seed all run on the GPU (sp_ecdsa_sign_rfc6979_batch)."""
    assert 0 <= msg_hash < 2**N_ELEMENT_BITS_ECDSA, "Message not signable."
    return batch.sign_many([msg_hash], [priv_key], [seed])[0]


def mimic_ec_mult_air(m: int, point: ECPoint, shift_point: ECPoint) -> ECPoint:
    """:176-190: m * point + shift_point the way the AIR does it (LSB first, always double, assert
    on every x-collision).  Python-int helper kept for API compatibility; verify() runs on the GPU."""
    assert 0 < m < 2**N_ELEMENT_BITS_ECDSA
    acc, base = shift_point, point
    for step in range(N_ELEMENT_BITS_ECDSA):
        assert acc[0] != base[0]
        if (m >> step) & 1:
            acc = ec_add(acc, base, FIELD_PRIME)
        base = ec_double(base, ALPHA, FIELD_PRIME)
    assert m >> N_ELEMENT_BITS_ECDSA == 0
    return acc


def is_point_on_curve(x: int, y: int) -> bool:
    """:193-194."""
    return pow(y, 2, FIELD_PRIME) == (pow(x, 3, FIELD_PRIME) + ALPHA * x + BETA) % FIELD_PRIME


def is_valid_stark_private_key(private_key: int) -> bool:
    """:197-201."""
    return 0 < private_key < EC_ORDER


def is_valid_stark_key(stark_key: int) -> bool:
    """:204-214."""
    y_squared = (stark_key * stark_key * stark_key + ALPHA * stark_key + BETA) % FIELD_PRIME
    return is_quad_residue(y_squared, FIELD_PRIME)


def verify(msg_hash: int, r: int, s: int, public_key: Union[int, ECPoint]) -> bool:
    """:217-260 through sp_ecdsa_verify_batch.  Pre-assert violations raise AssertionError with
    the reference's message; AIR-style failures and invalid x-only keys return False."""
    key = public_key if isinstance(public_key, int) else (public_key[0], public_key[1])
    code = batch.verify_codes([msg_hash], [r], [s], [key])[0]
    if code > batch.VERIFY_TRUE:
        batch.raise_for_verify_code(code, msg_hash, r, s)
    return code == batch.VERIFY_TRUE


def grind_key(key_seed: int, key_value_limit: int) -> int:
    """:263-288: sha256(seed || index) rejection sampling to a uniform value below the limit
    (host only; minimal-length big-endian encodings, at least one byte, like the JS twin)."""
    ceiling = (2**256 // key_value_limit) * key_value_limit  # largest multiple of the limit

    def minimal_bytes(value: int) -> bytes:
        return value.to_bytes(max(1, (value.bit_length() + 7) // 8), "big")

    prefix = minimal_bytes(key_seed)
    for index in itertools.count():
        candidate = int.from_bytes(hashlib.sha256(prefix + minimal_bytes(index)).digest(), "big")
        if candidate < ceiling:
            return candidate % key_value_limit


def pedersen_hash(*elements: int) -> int:
    """:296-297 (0, 1 or 2 elements).  One GPU launch through sp_pedersen_batch."""
    _check_elements(elements)
    x = elements[0] if len(elements) > 0 else 0
    y = elements[1] if len(elements) > 1 else 0
    return batch.pedersen_hash_many([x], [y])[0]


def pedersen_hash_as_point(*elements: int) -> ECPoint:
    """:300-318 (testing helper that also returns y)."""
    _check_elements(elements)
    x = elements[0] if len(elements) > 0 else 0
    y = elements[1] if len(elements) > 1 else 0
    return batch.pedersen_points_many([x], [y])[0]


def _check_elements(elements):
    for i, x in enumerate(elements):
        assert 0 <= x < FIELD_PRIME  # :307
        assert i < 2  # :311 - the table holds two elements' worth of points

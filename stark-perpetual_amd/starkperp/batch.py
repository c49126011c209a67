"""Batch entry points (additions next to the reference's scalar API): each is one call through
the C ABI and runs on the GPU.  Error behaviour follows the reference item by item:
an item that would trip an `assert` in signature.py raises AssertionError here."""
import ctypes

from . import _lib
from ._lib import new_bytes, new_felts, pack_felts, unpack_felts

FIELD_PRIME = 2**251 + 17 * 2**192 + 1
EC_ORDER = 0x800000000000010FFFFFFFFFFFFFFFFB781126DCAE7B2321E66A241ADC64D2F

HASH_OK, HASH_OUT_OF_RANGE, HASH_UNHASHABLE = 0, 1, 2


def _raise_hash_status(code):
    if code == HASH_OUT_OF_RANGE:
        raise AssertionError()  # signature.py:307  assert 0 <= x < FIELD_PRIME
    if code == HASH_UNHASHABLE:
        raise AssertionError("Unhashable input.")  # signature.py:313
    if code:
        raise AssertionError("pedersen status %d" % code)


def pedersen_hash_many(xs, ys):
    """[pedersen_hash(x, y) for x, y in zip(xs, ys)] (signature.py:296-318)."""
    n = len(xs)
    assert len(ys) == n
    if n == 0:
        return []
    for v in xs:
        assert 0 <= v < FIELD_PRIME
    for v in ys:
        assert 0 <= v < FIELD_PRIME
    lib = _lib.ensure_init()
    out, st = new_felts(n), new_bytes(n)
    _lib.check(lib.sp_pedersen_batch(pack_felts(xs), pack_felts(ys), out, st, n), "sp_pedersen_batch")
    for code in bytes(st)[:n]:
        if code:
            _raise_hash_status(code)
    return unpack_felts(out, n)


def pedersen_chain(elements):
    """Left fold H(...H(H(e0, e1), e2)..., ek) - the message / position hash-chain shape."""
    n = len(elements)
    assert n >= 1
    for v in elements:
        assert 0 <= v < FIELD_PRIME
    lib = _lib.ensure_init()
    out, st = new_felts(1), new_bytes(1)
    _lib.check(lib.sp_pedersen_chain(pack_felts(elements), n, out, st), "sp_pedersen_chain")
    _raise_hash_status(st[0] & 3 if st[0] in (0, 1, 2) else (2 if st[0] & 2 else 1))
    return unpack_felts(out, 1)[0]


def merkle_levels(leaves):
    """All levels (bottom-up) of the Pedersen Merkle tree over 2^h leaves."""
    n = len(leaves)
    assert n >= 1 and n & (n - 1) == 0
    for v in leaves:
        assert 0 <= v < FIELD_PRIME
    height = n.bit_length() - 1
    lib = _lib.ensure_init()
    root, st = new_felts(1), new_bytes(1)
    levels = new_felts(2 * n - 1)
    _lib.check(lib.sp_merkle_root(pack_felts(leaves), height, root, levels, st), "sp_merkle_root")
    if st[0]:
        _raise_hash_status(2 if st[0] & 2 else 1)
    flat = unpack_felts(levels, 2 * n - 1)
    out, pos, width = [], 0, n
    while width >= 1:
        out.append(flat[pos : pos + width])
        pos += width
        width //= 2
    return out


def merkle_root(leaves):
    n = len(leaves)
    assert n >= 1 and n & (n - 1) == 0
    for v in leaves:
        assert 0 <= v < FIELD_PRIME
    lib = _lib.ensure_init()
    root, st = new_felts(1), new_bytes(1)
    _lib.check(lib.sp_merkle_root(pack_felts(leaves), n.bit_length() - 1, root, None, st),
               "sp_merkle_root")
    if st[0]:
        _raise_hash_status(2 if st[0] & 2 else 1)
    return unpack_felts(root, 1)[0]

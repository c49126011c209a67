"""NumPy batch entry points: the same C-ABI calls as starkperp.batch with felts as `uint64[n, 4]`
little-endian limb arrays (the ABI's own layout), so that a 4096-order batch does not pay a Python
int <-> bytes conversion per field element (round 1: ~5 ms of the 17 ms host-inclusive C3 batch).
Range errors are reported the way the list API reports them (AssertionError for what signature.py
asserts); the arrays are handed to the library without a copy when they are C-contiguous uint64."""
import ctypes

import numpy as np

from . import _lib
from .batch import (EC_ORDER, FIELD_PRIME, HASH_OUT_OF_RANGE, HASH_UNHASHABLE, SIGN_BAD_INPUT, SIGN_OK, SIGN_RETRY,
                    VERIFY_TRUE, _raise_hash_status, raise_for_verify_code)

_P_LIMBS = np.array([(FIELD_PRIME >> (64 * i)) & (2**64 - 1) for i in range(4)], dtype=np.uint64)


def felts_from_ints(values) -> np.ndarray:
    """ints (0 <= v < 2^256) -> uint64[n, 4]."""
    raw = b"".join([int(v).to_bytes(32, "little") for v in values])
    return np.frombuffer(raw, dtype="<u8").reshape(len(values), 4).copy()


def ints_from_felts(arr) -> list:
    raw = np.ascontiguousarray(arr, dtype="<u8").tobytes()
    return [int.from_bytes(raw[32 * i : 32 * i + 32], "little") for i in range(len(raw) // 32)]


def _felts(a, n=None):
    a = np.ascontiguousarray(a, dtype=np.uint64)
    assert a.ndim == 2 and a.shape[1] == 4 and (n is None or a.shape[0] == n), "felts are uint64[n, 4]"
    return a


def _ptr(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def pack_fields(n, fields) -> np.ndarray:
    """Bit-packs 64-bit fields into felts: fields = [(uint64[n] values or an int, bit offset)], offsets
    non-overlapping - the word layouts of perpetual_messages.py without big-integer arithmetic."""
    out = np.zeros((n, 4), dtype=np.uint64)
    for values, offset in fields:
        v = np.broadcast_to(np.asarray(values, dtype=np.uint64), (n,))
        k, sh = divmod(offset, 64)
        out[:, k] |= v << np.uint64(sh)
        if sh and k + 1 < 4:
            out[:, k + 1] |= v >> np.uint64(64 - sh)
    return out


def pedersen_hash_many(x, y) -> np.ndarray:
    """uint64[n, 4] x, y -> uint64[n, 4] hashes (signature.py:296-318 per row)."""
    x = _felts(x)
    n = x.shape[0]
    y = _felts(y, n)
    out = np.empty((n, 4), dtype=np.uint64)
    if n == 0:
        return out
    st = np.zeros(n, dtype=np.uint8)
    _lib.check(_lib.ensure_init().sp_pedersen_batch(_ptr(x), _ptr(y), _ptr(out), _ptr(st), n), "sp_pedersen_batch")
    bad = np.flatnonzero(st)
    if bad.size:
        _raise_hash_status(int(st[bad[0]]))
    return out


def pedersen_chains(words) -> np.ndarray:
    """words uint64[depth, n, 4]: row i of the result = H(...H(H(w[0][i], w[1][i]), w[2][i])..., w[-1][i])."""
    w = np.ascontiguousarray(words, dtype=np.uint64)
    assert w.ndim == 3 and w.shape[2] == 4 and w.shape[0] >= 1
    depth, n = w.shape[0], w.shape[1]
    out = np.empty((n, 4), dtype=np.uint64)
    if n == 0:
        return out
    st = np.zeros(1, dtype=np.uint8)
    _lib.check(_lib.ensure_init().sp_pedersen_chains(_ptr(w), n, depth, _ptr(out), _ptr(st)), "sp_pedersen_chains")
    if st[0]:
        _raise_hash_status(HASH_UNHASHABLE if st[0] & 2 else HASH_OUT_OF_RANGE)
    return out


def verify_codes(z, r, s, qx, qy=None, key_tables=None) -> np.ndarray:
    """Result codes (include/starkperp.h SP_VERIFY_*) of verify(z, r, s, key) per row; qy None = x-only
    keys (signature.py:229-238).  key_tables as in starkperp.batch.verify_codes."""
    z = _felts(z)
    n = z.shape[0]
    r, s, qx = _felts(r, n), _felts(s, n), _felts(qx, n)
    qy = None if qy is None else _felts(qy, n)
    res = np.zeros(n, dtype=np.uint8)
    if n == 0:
        return res
    lib = _lib.ensure_init()
    fn = lib.sp_ecdsa_verify_batch if key_tables is None else (
        lib.sp_ecdsa_verify_batch_keyed if key_tables else None)
    if fn is None:
        raise ValueError("key_tables=False (forced ladder) is only offered by the list API")
    _lib.check(fn(_ptr(z), _ptr(r), _ptr(s), _ptr(qx), None if qy is None else _ptr(qy), _ptr(res), n),
               "sp_ecdsa_verify_batch")
    return res


def verify_many(z, r, s, qx, qy=None) -> np.ndarray:
    """bool[n]; raises the reference's AssertionError for the first row that violates a pre-assert."""
    codes = verify_codes(z, r, s, qx, qy)
    bad = np.flatnonzero(codes > VERIFY_TRUE)
    if bad.size:
        i = int(bad[0])
        zi, ri, si = (ints_from_felts(a[i : i + 1])[0] for a in (_felts(z), _felts(r), _felts(s)))
        raise_for_verify_code(int(codes[i]), zi, ri, si)
    return codes == VERIFY_TRUE


def sign_many(z, d, seeds=None):
    """sign(z, d, seed) per row (signature.py:137-173: RFC 6979 nonce, attempt and retry rule on the device,
    sp_ecdsa_sign_rfc6979_batch) without a Python int per field element: z, d uint64[n, 4], seeds uint64[n] or
    None (0 = no seed) -> (r, s) uint64[n, 4].  Raises the reference's AssertionError for a message >= 2^251
    and for a key outside [1, EC_ORDER) (starkperp.batch.sign_many); an item the device hands back after eight
    rejected nonces (a 2^-55 event each) is finished by the list API's host nonce generator."""
    z = _felts(z)
    n = z.shape[0]
    d = _felts(d, n)
    r, s = np.zeros((n, 4), dtype=np.uint64), np.zeros((n, 4), dtype=np.uint64)
    if n == 0:
        return r, s
    st = np.zeros(n, dtype=np.uint8)
    sd = None
    if seeds is not None:
        sd = np.ascontiguousarray(seeds, dtype=np.uint64)
        assert sd.shape == (n,), "seeds are uint64[n]"
    lib = _lib.ensure_init()
    _lib.check(lib.sp_ecdsa_sign_rfc6979_batch(_ptr(z), _ptr(d), None if sd is None else _ptr(sd), _ptr(r), _ptr(s),
                                               _ptr(st), n), "sp_ecdsa_sign_rfc6979_batch")
    bad = np.flatnonzero(st == SIGN_BAD_INPUT)
    if bad.size:
        i = int(bad[0])
        zi, di = ints_from_felts(z[i : i + 1])[0], ints_from_felts(d[i : i + 1])[0]
        assert 0 <= zi < 2**251, "Message not signable."
        raise AssertionError("private key must be in [1, EC_ORDER), got %s" % hex(di))
    left = np.flatnonzero(st == SIGN_RETRY)
    if left.size:
        from .batch import _sign_many_host_nonces
        zs, ds = ints_from_felts(z[left]), ints_from_felts(d[left])
        sl = [None if sd is None or int(sd[i]) == 0 else int(sd[i]) for i in left]
        sigs = _sign_many_host_nonces(zs, ds, sl)
        r[left] = felts_from_ints([a for a, _ in sigs])
        s[left] = felts_from_ints([b for _, b in sigs])
    return r, s


def order_ids(message_hashes) -> np.ndarray:
    """order/order.cairo:23-59: the 64 most significant bits of the 251-bit message hash, uint64[n]."""
    h = _felts(message_hashes)
    # 0 <= message_hash < SIGNED_MESSAGE_BOUND (order.cairo:22): a larger hash has no 64-bit order id
    assert not np.any(h[:, 3] >> np.uint64(59)), "message hash >= 2**251"
    return (h[:, 2] >> np.uint64(59)) | (h[:, 3] << np.uint64(5))


def limit_order_words(asset_sell, asset_buy, asset_fee, amount_sell, amount_buy, amount_fee, nonce, position_id,
                      expiration_timestamp) -> np.ndarray:
    """The five hash inputs of get_limit_order_msg_without_bounds (perpetual_messages.py:253-286) for n
    orders: uint64[5, n, 4].  Assets are felts (uint64[n, 4]); amounts / position ids uint64[n]; nonce and
    expiration below 2^32.  sell / buy are the caller's choice (is_buying_synthetic picks them)."""
    a_s, a_b, a_f = _felts(asset_sell), _felts(asset_buy), _felts(asset_fee)
    n = a_s.shape[0]
    u = lambda v: np.broadcast_to(np.asarray(v, dtype=np.uint64), (n,))
    assert int(u(nonce).max(initial=0)) < 2**32 and int(u(expiration_timestamp).max(initial=0)) < 2**32
    word0 = pack_fields(n, [(u(nonce), 0), (u(amount_fee), 32), (u(amount_buy), 96), (u(amount_sell), 160)])
    pos = u(position_id)
    word1 = pack_fields(n, [(u(expiration_timestamp), 17), (pos, 49), (pos, 113), (pos, 177), (3, 241)])
    return np.stack([a_s, a_b, a_f, word0, word1])


def limit_order_msgs(*args) -> np.ndarray:
    """Message hashes of n limit orders: four batched launches, arrays in and out."""
    return pedersen_chains(limit_order_words(*args))


TREE_NOT_COMMITTED = 0x80  # include/starkperp.h SP_TREE_NOT_COMMITTED


def order_batch(words, r, s, qx, tree, leaves, qy=None, id_shift=187):
    """BASELINE.json configs[2] in ONE library call (sp_order_batch): message-hash chains of the n orders
    (words uint64[depth, n, 4], e.g. limit_order_words; depth 1 = the message hashes themselves) -> verification of
    (z, r, s, key) through the key tables (z >= 2^251 is no signed message: verdict VERIFY_ASSERT_MSG = 5 and nothing
    is committed, constants.cairo:57) -> order ids (the top 64 bits of the 251-bit hash) -> update of `tree` (a state.LibrarySparseTree)
    with leaves[i] at order id i.  The verification overlaps the tree's level hashing on the device; the tree
    is committed only when every signature verified.
    Returns (z uint64[n, 4], verdict codes uint8[n], old_root, new_root, committed)."""
    w = np.ascontiguousarray(words, dtype=np.uint64)
    assert w.ndim == 3 and w.shape[2] == 4 and w.shape[0] >= 1
    depth, n = w.shape[0], w.shape[1]
    r, s, qx, leaves = _felts(r, n), _felts(s, n), _felts(qx, n), _felts(leaves, n)
    qy = None if qy is None else _felts(qy, n)
    z = np.empty((n, 4), dtype=np.uint64)
    verdicts = np.zeros(n, dtype=np.uint8)
    old, new, st = _lib.new_felts(1), _lib.new_felts(1), np.zeros(1, dtype=np.uint8)
    _lib.check(_lib.ensure_init().sp_order_batch(_ptr(w), depth, n, _ptr(r), _ptr(s), _ptr(qx),
                                                 None if qy is None else _ptr(qy), tree._handle, _ptr(leaves), id_shift,
                                                 _ptr(z), _ptr(verdicts), old, new, _ptr(st)), "sp_order_batch")
    if st[0] & (HASH_OUT_OF_RANGE | HASH_UNHASHABLE):
        _raise_hash_status(HASH_UNHASHABLE if st[0] & 2 else HASH_OUT_OF_RANGE)
    return z, verdicts, _lib.unpack_felts(old, 1)[0], _lib.unpack_felts(new, 1)[0], st[0] == 0

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


def pedersen_chain_right(elements):
    """Right fold H(e0, H(e1, ... H(e_{n-2}, e_{n-1}))) - cairo-lang's compute_hash_chain shape."""
    n = len(elements)
    assert n >= 1
    for v in elements:
        assert 0 <= v < FIELD_PRIME
    lib = _lib.ensure_init()
    out, st = new_felts(1), new_bytes(1)
    _lib.check(lib.sp_pedersen_chain_right(pack_felts(elements), n, out, st), "sp_pedersen_chain_right")
    if st[0]:
        _raise_hash_status(2 if st[0] & 2 else 1)
    return unpack_felts(out, 1)[0]


def pedersen_chains_many(chains):
    """Equal-depth chains: [H(...H(H(c[0], c[1]), c[2])..., c[-1]) for c in chains], evaluated
    level by level across the whole batch on the device (depth - 1 batched launches, one host
    round trip: sp_pedersen_chains)."""
    width = len(chains)
    if width == 0:
        return []
    depth = len(chains[0])
    assert depth >= 2 and all(len(c) == depth for c in chains)
    flat = []
    for column in zip(*chains):  # element j of chain i at index j * width + i
        assert min(column) >= 0 and max(column) < FIELD_PRIME
        flat.extend(column)
    lib = _lib.ensure_init()
    out, st = new_felts(width), new_bytes(1)
    _lib.check(lib.sp_pedersen_chains(pack_felts(flat), width, depth, out, st), "sp_pedersen_chains")
    if st[0]:
        _raise_hash_status(2 if st[0] & 2 else 1)
    return unpack_felts(out, width)


def pedersen_points_many(xs, ys):
    """[pedersen_hash_as_point(x, y) ...] (signature.py:300-318) - the full affine point."""
    n = len(xs)
    assert len(ys) == n
    if n == 0:
        return []
    for v in list(xs) + list(ys):
        assert 0 <= v < FIELD_PRIME
    lib = _lib.ensure_init()
    ox, oy, st = new_felts(n), new_felts(n), new_bytes(n)
    _lib.check(lib.sp_pedersen_point_batch(pack_felts(xs), pack_felts(ys), ox, oy, st, n),
               "sp_pedersen_point_batch")
    for code in bytes(st)[:n]:
        if code:
            _raise_hash_status(code)
    return list(zip(unpack_felts(ox, n), unpack_felts(oy, n)))


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


def merkle_roots_many(trees):
    """Roots of any number of independent trees of equal power-of-two size, built in lockstep
    (sp_merkle_forest_dev: one launch pair per level for all trees)."""
    import torch  # device staging for the forest buffer
    count = len(trees)
    assert count >= 1
    n = len(trees[0])
    assert n >= 1 and n & (n - 1) == 0 and all(len(t) == n for t in trees)
    height = n.bit_length() - 1
    flat = [v for t in trees for v in t]
    for v in flat:
        assert 0 <= v < FIELD_PRIME
    lib = _lib.ensure_init()
    total = count * (2 * n - 1)
    import numpy as np
    raw = b"".join(int(v).to_bytes(32, "little") for v in flat)
    buf = torch.zeros((total, 4), dtype=torch.int64, device="cuda")
    buf[: len(flat)] = torch.from_numpy(np.frombuffer(raw, dtype="<i8").reshape(len(flat), 4).copy()).cuda()
    st = new_bytes(1)
    _lib.check(lib.sp_merkle_forest_dev(buf.data_ptr(), count, height, st,
                                        torch.cuda.current_stream().cuda_stream), "sp_merkle_forest_dev")
    torch.cuda.synchronize()
    if st[0]:
        _raise_hash_status(2 if st[0] & 2 else 1)
    rootraw = buf[total - count :].cpu().numpy().astype("<i8").tobytes()
    return [int.from_bytes(rootraw[32 * i : 32 * i + 32], "little") for i in range(count)]


def merkle_sparse_root(height, modifications, empty_leaf=0):
    """Root of the height-`height` tree (<= 64) holding {index: leaf} and `empty_leaf` elsewhere:
    the multi-update walk of starkware/python/merkle_tree.py:4-26 with every hash on the GPU."""
    items = sorted(dict(modifications).items())
    n = len(items)
    for k, v in items:
        assert 0 <= k < (1 << height) and 0 <= v < FIELD_PRIME
    assert 0 <= empty_leaf < FIELD_PRIME
    lib = _lib.ensure_init()
    keys = (ctypes.c_uint64 * max(n, 1))(*[k for k, _ in items])
    root, st = new_felts(1), new_bytes(1)
    _lib.check(lib.sp_merkle_sparse_root(keys, pack_felts([v for _, v in items]), n, height,
                                         pack_felts([empty_leaf]), root, st), "sp_merkle_sparse_root")
    if st[0]:
        _raise_hash_status(2 if st[0] & 2 else 1)
    return unpack_felts(root, 1)[0]


# ---- ECDSA ------------------------------------------------------------------------------------
VERIFY_FALSE, VERIFY_TRUE = 0, 1
VERIFY_ASSERT_S, VERIFY_ASSERT_R, VERIFY_ASSERT_W, VERIFY_ASSERT_MSG, VERIFY_ASSERT_CURVE = 2, 3, 4, 5, 6
VERIFY_STALE_SLOT = 7  # sp_ecdsa_verify_keyed_dev: slot handle from before a key-cache reset
SIGN_OK, SIGN_RETRY, SIGN_BAD_INPUT = 0, 1, 2
_TWO251 = 2**251


def _verify_inputs(msg_hashes, rs, ss, public_keys):
    n = len(msg_hashes)
    assert len(rs) == len(ss) == len(public_keys) == n
    xonly = isinstance(public_keys[0], int)
    assert all(isinstance(q, int) == xonly for q in public_keys), "mix of x-only and point keys"
    p = FIELD_PRIME
    if xonly:
        qx, qy = [q if 0 <= q < p else q % p for q in public_keys], None
    else:
        qx = [q[0] if 0 <= q[0] < p else q[0] % p for q in public_keys]
        qy = pack_felts([q[1] if 0 <= q[1] < p else q[1] % p for q in public_keys])
    # Values that do not fit the 256-bit ABI fail the same pre-asserts as in signature.py:219-227;
    # they are clamped to an out-of-range representative so the kernel reports the right code.
    top = 2**256
    z = [v if 0 <= v < top else top - 1 for v in msg_hashes]
    r = [v if 0 <= v < top else 0 for v in rs]
    s = [v if 0 <= v < top else 0 for v in ss]
    return pack_felts(z), pack_felts(r), pack_felts(s), pack_felts(qx), qy


def verify_codes(msg_hashes, rs, ss, public_keys, key_tables=None):
    """Raw per-item result codes of sp_ecdsa_verify_batch (include/starkperp.h SP_VERIFY_*).
    public_keys: all ints (x-only, signature.py:229-238) or all (x, y) pairs.
    key_tables: None = the library's policy (per-key comb tables when most keys of the batch are
    already registered), True = always through the tables, False = always the per-signature
    ladder.  The result is the same either way."""
    n = len(msg_hashes)
    if n == 0:
        return []
    z, r, s, qx, qy = _verify_inputs(msg_hashes, rs, ss, public_keys)
    lib = _lib.ensure_init()
    res = new_bytes(n)
    if key_tables is None:
        _lib.check(lib.sp_ecdsa_verify_batch(z, r, s, qx, qy, res, n), "sp_ecdsa_verify_batch")
    elif key_tables:
        _lib.check(lib.sp_ecdsa_verify_batch_keyed(z, r, s, qx, qy, res, n), "sp_ecdsa_verify_batch_keyed")
    else:  # stage by hand so that the policy of the host entry point is bypassed
        import ctypes
        import torch
        dev = [torch.frombuffer(bytearray(bytes(b)), dtype=torch.int64).cuda() if b is not None else None
               for b in (z, r, s, qx, qy)]
        out = torch.zeros(n, dtype=torch.uint8, device="cuda")
        _lib.check(lib.sp_ecdsa_verify_batch_dev(dev[0].data_ptr(), dev[1].data_ptr(), dev[2].data_ptr(),
                                                 dev[3].data_ptr(), dev[4].data_ptr() if dev[4] is not None else None,
                                                 out.data_ptr(), n, torch.cuda.current_stream().cuda_stream),
                   "sp_ecdsa_verify_batch_dev")
        torch.cuda.synchronize()
        return out.cpu().tolist()
    return list(bytes(res)[:n])


def register_keys(public_keys):
    """Registers public keys (all ints = x-only, or all (x, y) pairs) in the library's key-table
    cache; returns one slot index per key (equal keys share a slot)."""
    import ctypes
    n = len(public_keys)
    if n == 0:
        return []
    xonly = isinstance(public_keys[0], int)
    assert all(isinstance(q, int) == xonly for q in public_keys), "mix of x-only and point keys"
    qx = pack_felts([int(q if xonly else q[0]) % FIELD_PRIME for q in public_keys])
    qy = None if xonly else pack_felts([int(q[1]) % FIELD_PRIME for q in public_keys])
    slots = (ctypes.c_uint32 * n)()
    _lib.check(_lib.ensure_init().sp_ecdsa_register_keys(qx, qy, n, slots), "sp_ecdsa_register_keys")
    return list(slots)


def key_cache_info():
    import ctypes
    cap, used = ctypes.c_size_t(), ctypes.c_size_t()
    _lib.check(_lib.ensure_init().sp_ecdsa_key_cache_info(ctypes.byref(cap), ctypes.byref(used)),
               "sp_ecdsa_key_cache_info")
    return cap.value, used.value


def key_cache_reset():
    _lib.check(_lib.ensure_init().sp_ecdsa_key_cache_reset(), "sp_ecdsa_key_cache_reset")


VERIFY_POLICY_AUTO, VERIFY_POLICY_LADDER, VERIFY_POLICY_KEYED = 0, 1, 2


def set_verify_policy(policy: int):
    """How verify_many / sp_ecdsa_verify_batch choose between the per-signature ladder and the key tables
    (include/starkperp.h): AUTO remembers keys between calls, LADDER is stateless, KEYED always registers."""
    _lib.check(_lib.ensure_init().sp_ecdsa_set_verify_policy(int(policy)), "sp_ecdsa_set_verify_policy")


def get_verify_policy() -> int:
    return int(_lib.ensure_init().sp_ecdsa_get_verify_policy())


def raise_for_verify_code(code, msg_hash, r, s):
    """Re-creates the reference's assertion (text included) for a pre-assert code."""
    if code == VERIFY_ASSERT_S:
        raise AssertionError("s = %s" % s)
    if code == VERIFY_ASSERT_R:
        raise AssertionError("r = %s" % r)
    if code == VERIFY_ASSERT_W:
        raise AssertionError("w = %s" % pow(s, -1, EC_ORDER))
    if code == VERIFY_ASSERT_MSG:
        raise AssertionError("msg_hash = %s" % msg_hash)
    if code == VERIFY_ASSERT_CURVE:
        raise AssertionError()
    if code == VERIFY_STALE_SLOT:
        raise _lib.StarkPerpError("key slot handle is stale (from before a key-cache reset) or was never handed out")


def verify_many(msg_hashes, rs, ss, public_keys):
    """[verify(z, r, s, q) ...] (signature.py:217-260); raises AssertionError for the first item
    whose inputs violate a pre-assert, like the scalar function would."""
    codes = verify_codes(msg_hashes, rs, ss, public_keys)
    for i, code in enumerate(codes):
        if code > VERIFY_TRUE:
            raise_for_verify_code(code, msg_hashes[i], rs[i], ss[i])
    return [c == VERIFY_TRUE for c in codes]


def public_keys_many(priv_keys):
    """[private_key_to_ec_point_on_stark_curve(d) ...] (signature.py:104-106)."""
    n = len(priv_keys)
    if n == 0:
        return []
    for d in priv_keys:
        assert 0 < d < EC_ORDER
    lib = _lib.ensure_init()
    qx, qy, st = new_felts(n), new_felts(n), new_bytes(n)
    _lib.check(lib.sp_public_key_batch(pack_felts(priv_keys), qx, qy, st, n), "sp_public_key_batch")
    assert not any(bytes(st)[:n])
    return list(zip(unpack_felts(qx, n), unpack_felts(qy, n)))


def sign_attempt_many(msg_hashes, priv_keys, ks):
    """One pass of the loop body of sign() (signature.py:146-173) per item with explicit nonces.
    Returns (rs, ss, status)."""
    n = len(msg_hashes)
    lib = _lib.ensure_init()
    r, s, st = new_felts(n), new_felts(n), new_bytes(n)
    _lib.check(lib.sp_ecdsa_sign_batch(pack_felts(msg_hashes), pack_felts(priv_keys), pack_felts(ks),
                                       r, s, st, n), "sp_ecdsa_sign_batch")
    return unpack_felts(r, n), unpack_felts(s, n), list(bytes(st)[:n])


def sign_dev(z, d, seeds=None, k=None, stream=None):
    """Signing on tensors that already live in HBM (sp_ecdsa_sign_rfc6979_batch_dev; with caller nonces `k`
    sp_ecdsa_sign_batch_dev: one attempt, signature.py:146-173): z, d (and k) int64[n, 4] felts on the GPU, seeds
    int64[n] or None -> (r, s, status) tensors on the same device, status uint8[n] of SIGN_OK / SIGN_RETRY /
    SIGN_BAD_INPUT (include/starkperp.h).  One launch on the current stream, nothing is copied or waited for; r / s
    of an item that is not SIGN_OK stay zero."""
    import torch
    n = z.shape[0]
    assert z.is_cuda and z.dtype == torch.int64 and z.shape == (n, 4) and z.is_contiguous(), "z: int64[n, 4] on the GPU"
    for t in (d, k):
        assert t is None or (t.device == z.device and t.dtype == torch.int64 and t.shape == (n, 4) and t.is_contiguous())
    assert seeds is None or k is None, "a caller nonce leaves no room for a seed"
    if seeds is not None:
        assert seeds.device == z.device and seeds.dtype == torch.int64 and seeds.shape == (n,) and seeds.is_contiguous()
    r, s = torch.zeros_like(z), torch.zeros_like(z)
    st = torch.zeros(n, dtype=torch.uint8, device=z.device)
    if n == 0:
        return r, s, st
    lib = _lib.ensure_init()
    h = torch.cuda.current_stream(z.device).cuda_stream if stream is None else stream
    if k is None:
        _lib.check(lib.sp_ecdsa_sign_rfc6979_batch_dev(z.data_ptr(), d.data_ptr(),
                                                       None if seeds is None else seeds.data_ptr(), r.data_ptr(),
                                                       s.data_ptr(), st.data_ptr(), n, h), "sp_ecdsa_sign_rfc6979_batch_dev")
    else:
        _lib.check(lib.sp_ecdsa_sign_batch_dev(z.data_ptr(), d.data_ptr(), k.data_ptr(), r.data_ptr(), s.data_ptr(),
                                               st.data_ptr(), n, h), "sp_ecdsa_sign_batch_dev")
    return r, s, st


def public_keys_dev(d, want_y=True, stream=None):
    """d int64[n, 4] on the GPU -> (qx, qy or None, status) on the same device (sp_public_key_batch_dev,
    signature.py:104-106); status 0 ok, SIGN_BAD_INPUT for d outside (0, EC_ORDER) - that row stays zero."""
    import torch
    n = d.shape[0]
    assert d.is_cuda and d.dtype == torch.int64 and d.shape == (n, 4) and d.is_contiguous(), "d: int64[n, 4] on the GPU"
    qx = torch.zeros_like(d)
    qy = torch.zeros_like(d) if want_y else None
    st = torch.zeros(n, dtype=torch.uint8, device=d.device)
    if n:
        lib = _lib.ensure_init()
        h = torch.cuda.current_stream(d.device).cuda_stream if stream is None else stream
        _lib.check(lib.sp_public_key_batch_dev(d.data_ptr(), qx.data_ptr(), None if qy is None else qy.data_ptr(),
                                               st.data_ptr(), n, h), "sp_public_key_batch_dev")
    return qx, qy, st


def sign_many(msg_hashes, priv_keys, seeds=None):
    """[sign(z, d, seed) ...] (signature.py:137-173) in one launch: RFC 6979 nonce, k*G, the mod-N
    finish and the reference's retry rule all run on the GPU (sp_ecdsa_sign_rfc6979_batch).  Items
    the device hands back (a seed that does not fit 64 bits, or eight rejected nonces in a row)
    go through the host nonce generator instead, attempt by attempt."""
    n = len(msg_hashes)
    assert len(priv_keys) == n
    seeds = [None] * n if seeds is None else list(seeds)
    for z in msg_hashes:
        assert 0 <= z < _TWO251, "Message not signable."
    # The reference's sign() has no range check on the private key (signature.py:137-173); keys outside
    # [1, EC_ORDER) are meaningless there (0 signs with the point at infinity, larger values alias
    # key mod N with a different nonce).  Here they are rejected up front, never reduced silently.
    for d in priv_keys:
        assert 0 < d < EC_ORDER, "private key must be in [1, EC_ORDER), got %s" % hex(d)
    if n == 0:
        return []
    out = [None] * n
    on_device = [i for i in range(n) if seeds[i] is None or 0 <= seeds[i] < 2**64]
    left = [i for i in range(n) if not (seeds[i] is None or 0 <= seeds[i] < 2**64)]
    if on_device:
        m = len(on_device)
        lib = _lib.ensure_init()
        r, s, st = new_felts(m), new_felts(m), new_bytes(m)
        seed_arr = (ctypes.c_uint64 * m)(*[seeds[i] or 0 for i in on_device])
        _lib.check(lib.sp_ecdsa_sign_rfc6979_batch(pack_felts([msg_hashes[i] for i in on_device]),
                                                   pack_felts([priv_keys[i] for i in on_device]),
                                                   seed_arr, r, s, st, m), "sp_ecdsa_sign_rfc6979_batch")
        rs, ss = unpack_felts(r, m), unpack_felts(s, m)
        for j, i in enumerate(on_device):
            if st[j] == SIGN_OK:
                out[i] = (rs[j], ss[j])
            elif st[j] == SIGN_RETRY:
                left.append(i)
            else:
                raise AssertionError("sign: input out of range")
    if left:
        for i, sig in zip(left, _sign_many_host_nonces([msg_hashes[i] for i in left], [priv_keys[i] for i in left],
                                                       [seeds[i] for i in left])):
            out[i] = sig
    return out


def _sign_many_host_nonces(msg_hashes, priv_keys, seeds):
    """The same loop with RFC 6979 on the host (starkperp/rfc6979.py) and one GPU attempt per nonce."""
    from .signature import generate_k_rfc6979  # late import (signature imports batch)

    n = len(msg_hashes)
    seeds = list(seeds)
    out = [None] * n
    todo = list(range(n))
    while todo:
        ks = [generate_k_rfc6979(msg_hashes[i], priv_keys[i], seeds[i]) for i in todo]
        for i in todo:
            seeds[i] = 1 if seeds[i] is None else seeds[i] + 1
        rs, ss, st = sign_attempt_many([msg_hashes[i] for i in todo], [priv_keys[i] for i in todo], ks)
        again = []
        for j, i in enumerate(todo):
            if st[j] == SIGN_OK:
                out[i] = (rs[j], ss[j])
            elif st[j] == SIGN_RETRY:
                again.append(i)
            else:
                raise AssertionError("sign: input out of range")
        todo = again
    return out

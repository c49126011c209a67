"""Deterministic nonce generation (RFC 6979 section 3.2, HMAC-SHA256) with the conventions of
python-ecdsa 0.17's `ecdsa.rfc6979.generate_k`, the third-party routine the reference calls at
signature.py:25,128-134.  Host-side byte work; the elliptic-curve part of signing is on the GPU."""
import hashlib
import hmac


def _bits2int(data: bytes, qlen: int) -> int:
    value = int.from_bytes(data, "big")
    surplus = 8 * len(data) - qlen
    if surplus > 0:
        value >>= surplus
    return value


def generate_k(order: int, secexp: int, hash_func, data: bytes, retry_gen: int = 0,
               extra_entropy: bytes = b"") -> int:
    qlen = order.bit_length()
    holen = hash_func().digest_size
    rolen = (qlen + 7) // 8
    order_len = (len("%x" % order) + 1) // 2
    h1 = _bits2int(data, qlen)
    if h1 >= order:
        h1 -= order
    key_material = secexp.to_bytes(order_len, "big") + h1.to_bytes(order_len, "big") + extra_entropy

    def prf(key, *parts):
        return hmac.new(key, b"".join(parts), hash_func).digest()

    v = b"\x01" * holen
    k = prf(b"\x00" * holen, v, b"\x00", key_material)
    v = prf(k, v)
    k = prf(k, v, b"\x01", key_material)
    v = prf(k, v)
    while True:
        t = b""
        while len(t) < rolen:
            v = prf(k, v)
            t += v
        candidate = _bits2int(t, qlen)
        if 1 <= candidate < order:
            if retry_gen <= 0:
                return candidate
            retry_gen -= 1
        k = prf(k, v, b"\x00")
        v = prf(k, v)

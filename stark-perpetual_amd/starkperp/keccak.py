"""Keccak-256 (the pre-NIST padding Ethereum uses; hashlib's sha3_256 pads differently), host side.

Only `build_condition` needs it (reference: services/perpetual/public/perpetual_messages.py:15-21
calls `Web3.solidityKeccak(["address", "bytes32"], ...)`; web3 is an external dependency that is
not part of the reference tree).  Restated from the Keccak reference specification: 1600-bit state,
rate 1088, 24 rounds, multi-rate padding 0x01 ... 0x80.  Pinned by the published test vectors in
tests/test_host_messages.py (empty string, "abc") and, for multi-block absorption, by running the
same sponge with SHA3's domain byte against hashlib.sha3_256."""

_MASK = (1 << 64) - 1
_ROUNDS = 24
_RATE = 136


def _round_constants():
    out, r = [], 1
    for _ in range(_ROUNDS):
        rc = 0
        for j in range(7):
            r = ((r << 1) ^ ((r >> 7) * 0x71)) & 0xFF
            if r & 2:
                rc ^= 1 << ((1 << j) - 1)
        out.append(rc)
    return out


def _rotation_offsets():
    rot = [[0] * 5 for _ in range(5)]
    x, y = 1, 0
    for t in range(24):
        rot[x][y] = ((t + 1) * (t + 2) // 2) % 64
        x, y = y, (2 * x + 3 * y) % 5
    return rot


_RC = _round_constants()
_ROT = _rotation_offsets()


def _rol(v, n):
    return ((v << n) | (v >> (64 - n))) & _MASK if n else v


def _permute(a):
    for rc in _RC:
        c = [a[x][0] ^ a[x][1] ^ a[x][2] ^ a[x][3] ^ a[x][4] for x in range(5)]
        d = [c[(x - 1) % 5] ^ _rol(c[(x + 1) % 5], 1) for x in range(5)]
        a = [[a[x][y] ^ d[x] for y in range(5)] for x in range(5)]
        b = [[0] * 5 for _ in range(5)]
        for x in range(5):
            for y in range(5):
                b[y][(2 * x + 3 * y) % 5] = _rol(a[x][y], _ROT[x][y])
        a = [[b[x][y] ^ (~b[(x + 1) % 5][y] & _MASK & b[(x + 2) % 5][y]) for y in range(5)] for x in range(5)]
        a[0][0] ^= rc
    return a


def sponge256(data: bytes, domain: int) -> bytes:
    """Rate-1088 sponge with a 256-bit output; domain byte 0x01 = Keccak-256, 0x06 = SHA3-256 (the
    latter only so that tests can cross-check the permutation against hashlib)."""
    padded = bytearray(data)
    padded.append(domain)
    padded.extend(b"\x00" * (-len(padded) % _RATE))
    padded[-1] |= 0x80
    state = [[0] * 5 for _ in range(5)]
    for off in range(0, len(padded), _RATE):
        block = padded[off : off + _RATE]
        for i in range(_RATE // 8):
            state[i % 5][i // 5] ^= int.from_bytes(block[8 * i : 8 * i + 8], "little")
        state = _permute(state)
    return b"".join(state[i % 5][i // 5].to_bytes(8, "little") for i in range(4))


def keccak256(data: bytes) -> bytes:
    return sponge256(data, 0x01)

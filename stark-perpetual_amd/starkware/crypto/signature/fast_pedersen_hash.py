"""starkware.crypto.signature.fast_pedersen_hash API (fast_pedersen_hash.py:34-52) on the GPU:
`pedersen_hash(x, y) -> int` and the bytes32 variant `pedersen_hash_func`."""
from starkperp.signature import pedersen_hash as _pedersen_hash

LOW_PART_BITS = 248  # fast_pedersen_hash.py:17-18 (the 248 + 4 bit split of each element)
LOW_PART_MASK = 2**248 - 1


def pedersen_hash(x: int, y: int) -> int:
    return _pedersen_hash(x, y)


def pedersen_hash_func(x: bytes, y: bytes) -> bytes:
    assert len(x) == len(y) == 32, "Unexpected element length."
    return _pedersen_hash(int.from_bytes(x, "big"), int.from_bytes(y, "big")).to_bytes(32, "big")

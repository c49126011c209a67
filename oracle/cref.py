"""ctypes wrapper of oracle/starkref.c (ORACLE - test infrastructure only).  Builds the shared
library on first use (gcc -O3 -fopenmp) into oracle/_build/ (git-ignored)."""
import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libstarkref.so")
_SRC = os.path.join(_HERE, "starkref.c")
_lib = None


def build():
    os.makedirs(os.path.dirname(_SO), exist_ok=True)
    subprocess.check_call(["gcc", "-O3", "-fopenmp", "-shared", "-fPIC", _SRC, "-o", _SO])


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(_SRC):
            build()
        _lib = ctypes.CDLL(_SO)
        _lib.cref_max_threads.restype = ctypes.c_int
    return _lib


def _pack(values):
    raw = b"".join(int(v).to_bytes(32, "little") for v in values)
    return (ctypes.c_uint64 * (4 * len(values))).from_buffer_copy(raw)


def _unpack(buf, n):
    raw = bytes(buf)
    return [int.from_bytes(raw[32 * i : 32 * i + 32], "little") for i in range(n)]


def max_threads():
    return int(lib().cref_max_threads())


def pedersen_hash_many(xs, ys):
    """Returns (outputs, status bytes): the 252-step affine loop of signature.py:300-318 in C."""
    n = len(xs)
    out = (ctypes.c_uint64 * (4 * n))()
    st = (ctypes.c_uint8 * n)()
    lib().cref_pedersen_batch(_pack(xs), _pack(ys), out, st, ctypes.c_size_t(n))
    return _unpack(out, n), list(bytes(st))


def pedersen_chain_right(elements):
    """H(e0, H(e1, ... H(e_{n-2}, e_{n-1}))) in one C call (12 000 links in about a second)."""
    n = len(elements)
    out = (ctypes.c_uint64 * 4)()
    status = lib().cref_pedersen_chain_right(_pack(elements), ctypes.c_size_t(n), out)
    assert status == 0, "unhashable link"
    return _unpack(out, 1)[0]


def opt_pedersen_hash_many(xs, ys):
    """The optimised comparator (windowed tables + batched affine additions): same outputs."""
    n = len(xs)
    out = (ctypes.c_uint64 * (4 * n))()
    st = (ctypes.c_uint8 * n)()
    lib().cref_opt_pedersen_batch(_pack(xs), _pack(ys), out, st, ctypes.c_size_t(n))
    return _unpack(out, n), list(bytes(st))


def opt_merkle_levels(leaves):
    n = len(leaves)
    height = n.bit_length() - 1
    assert n == 1 << height
    buf = (ctypes.c_uint64 * (4 * (2 * n - 1)))()
    ctypes.memmove(buf, _pack(leaves), 32 * n)
    lib().cref_opt_merkle_build(buf, ctypes.c_uint(height))
    flat = _unpack(buf, 2 * n - 1)
    levels, off, width = [], 0, n
    while width >= 1:
        levels.append(flat[off : off + width])
        off += width
        width //= 2
    return levels


def opt_merkle_timed(leaves, min_seconds=2.0, max_reps=256):
    """Repeats the optimised rebuild on one marshalled buffer (the leaves stay in place, the upper levels
    are overwritten): returns (levels, repetitions, seconds inside the C library)."""
    import time
    n = len(leaves)
    height = n.bit_length() - 1
    assert n == 1 << height
    buf = (ctypes.c_uint64 * (4 * (2 * n - 1)))()
    ctypes.memmove(buf, _pack(leaves), 32 * n)
    lib().cref_opt_merkle_build(buf, ctypes.c_uint(height))  # builds the window table on first use
    reps, t0, dt = 0, time.time(), 0.0
    while dt < min_seconds and reps < max_reps:
        lib().cref_opt_merkle_build(buf, ctypes.c_uint(height))
        reps += 1
        dt = time.time() - t0
    flat = _unpack(buf, 2 * n - 1)
    levels, off, width = [], 0, n
    while width >= 1:
        levels.append(flat[off : off + width])
        off += width
        width //= 2
    return levels, reps, dt


def merkle_levels(leaves):
    n = len(leaves)
    height = n.bit_length() - 1
    buf = (ctypes.c_uint64 * (4 * (2 * n - 1)))()
    ctypes.memmove(buf, _pack(leaves), 32 * n)
    lib().cref_merkle_build(buf, ctypes.c_uint(height))
    flat = _unpack(buf, 2 * n - 1)
    out, pos, width = [], 0, n
    while width >= 1:
        out.append(flat[pos : pos + width])
        pos += width
        width //= 2
    return out


def public_keys_many(ds):
    n = len(ds)
    qx, qy = (ctypes.c_uint64 * (4 * n))(), (ctypes.c_uint64 * (4 * n))()
    lib().cref_public_key_batch(_pack(ds), qx, qy, ctypes.c_size_t(n))
    return list(zip(_unpack(qx, n), _unpack(qy, n)))


def verify_codes(zs, rs, ss, points):
    """Point keys only; codes: 0 False, 1 True, 2..6 = which pre-assert (include/starkperp.h)."""
    n = len(zs)
    res = (ctypes.c_uint8 * n)()
    lib().cref_verify_batch(_pack(zs), _pack(rs), _pack(ss), _pack([p[0] for p in points]),
                            _pack([p[1] for p in points]), res, ctypes.c_size_t(n))
    return list(bytes(res))

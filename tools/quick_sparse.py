#!/usr/bin/env python3
"""Sparse multi-update timing (development aid): python tools/quick_sparse.py [n_leaves=4096] [height=64]"""
import os, sys, time, random, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "stark-perpetual_amd"))
from starkperp import _lib, batch
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
height = int(sys.argv[2]) if len(sys.argv) > 2 else 64
rng = random.Random(3)
keys = sorted(set(rng.randrange(2**height) for _ in range(n)))
leaves = [rng.randrange(batch.FIELD_PRIME) for _ in keys]
lib = _lib.ensure_init()
k = (ctypes.c_uint64 * len(keys))(*keys)
lv = _lib.pack_felts(leaves)
empty = _lib.pack_felts([0])
root = _lib.new_felts(1)
st = (ctypes.c_uint8 * 1)()
for it in range(4):
    t0 = time.perf_counter()
    _lib.check(lib.sp_merkle_sparse_root(k, lv, len(keys), height, empty, root, st), "sparse")
    print("call %d: %.3f ms" % (it, (time.perf_counter() - t0) * 1e3))

#!/usr/bin/env python3
"""Time sp_tree_update alone (arrays pre-packed) and show where a height-64 update goes (dev aid)."""
import os, sys, time, random, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "stark-perpetual_amd"))
from starkperp import _lib
lib = _lib.ensure_init(0, 26)
rng = random.Random(5)
n, height = 4096, 64
P = 2**251 + 17 * 2**192 + 1
h = ctypes.c_int()
_lib.check(lib.sp_tree_create(height, _lib.pack_felts([0]), ctypes.byref(h)), "create")
for it in range(4):
    mods = sorted({rng.randrange(2**height): rng.randrange(P) for _ in range(n)}.items())
    t0 = time.perf_counter()
    keys = (ctypes.c_uint64 * n)(*[k for k, _ in mods])
    vals = _lib.pack_felts([v for _, v in mods])
    t1 = time.perf_counter()
    old, new, st = _lib.new_felts(1), _lib.new_felts(1), _lib.new_bytes(1)
    _lib.check(lib.sp_tree_update(h.value, keys, vals, n, old, new, st), "update")
    t2 = time.perf_counter()
    print("batch %d: pack %.2f ms, sp_tree_update %.2f ms" % (it, (t1 - t0) * 1e3, (t2 - t1) * 1e3))

#!/usr/bin/env python3
"""Time of sp_init (table build) for one window plan: python tools/quick_init.py [window_bits=26]
STARKPERP_TABLE_BUILD=direct selects the round 1 - 3 build (every entry from its set bits)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "stark-perpetual_amd"))
import torch
torch.zeros(1, device="cuda")
from starkperp import _lib
wb = int(sys.argv[1]) if len(sys.argv) > 1 else 26
lib = _lib.load()
for rep in range(2):
    t = time.perf_counter()
    _lib.check(lib.sp_init(0, wb), "sp_init")
    dt = time.perf_counter() - t
    print("build=%s window_bits=%d tables=%.1f GiB: sp_init %.3f s" % (
        os.environ.get("STARKPERP_TABLE_BUILD", "doubling"), lib.sp_window_bits(), lib.sp_table_bytes() / 2**30, dt))
    lib.sp_shutdown()

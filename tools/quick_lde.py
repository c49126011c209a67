#!/usr/bin/env python3
"""LDE only (4 columns 2^20 -> 2^22), for kernel traces of the NTT passes (dev aid)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "stark-perpetual_amd"))
import torch
from starkperp import _lib, stark
lib = _lib.ensure_init(0, 16)
g = torch.Generator().manual_seed(1)
n = 1 << 20
t = torch.randint(-(2**63), 2**63 - 1, (4, n, 4), dtype=torch.int64, generator=g); t[:, :, 3] &= (1 << 58) - 1
t = t.cuda()
stark.lde(t); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5): stark.lde(t)
torch.cuda.synchronize()
print("lde 4 cols 2^20 -> 2^22: %.3f ms" % ((time.perf_counter() - t0) / 5 * 1e3))

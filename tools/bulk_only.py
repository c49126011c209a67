#!/usr/bin/env python3
"""Bulk Pedersen batch only (profiling target): python tools/bulk_only.py [log2_n] [window_bits]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "stark-perpetual_amd"))
import torch
from starkperp import _lib
logn = int(sys.argv[1]) if len(sys.argv) > 1 else 22
wb = int(sys.argv[2]) if len(sys.argv) > 2 else 0
lib = _lib.ensure_init(0, wb)
n = 1 << logn
g = torch.Generator().manual_seed(1)
def felts(seed):
    t = torch.randint(-(2**63), 2**63 - 1, (n, 4), dtype=torch.int64, generator=g); t[:, 3] &= (1 << 58) - 1
    return t.cuda()
x, y = felts(1), felts(2); o = torch.empty_like(x)
s = torch.cuda.current_stream().cuda_stream
for _ in range(4):
    _lib.check(lib.sp_pedersen_batch_dev(x.data_ptr(), y.data_ptr(), o.data_ptr(), None, n, s), "ped")
torch.cuda.synchronize()

#!/usr/bin/env python3
"""Bulk Pedersen rate for one window plan (development aid): python tools/quick_bulk.py [log2_n=22] [window_bits=0] [same]"""
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
def felts():
    t = torch.randint(-(2**63), 2**63 - 1, (n, 4), dtype=torch.int64, generator=g); t[:, 3] &= (1 << 58) - 1
    return t.cuda()
x, y = felts(), felts(); o = torch.empty_like(x)
if len(sys.argv) > 3 and sys.argv[3] == "same":  # every hash the same pair: all gathers hit 19 cached entries (the kernel without its memory side)
    x[:] = x[0]; y[:] = y[0]
if len(sys.argv) > 3 and sys.argv[3].startswith("few"):  # few<k>: 2^k distinct pairs repeated: random-looking lanes, gathers served by L2
    m = 1 << int(sys.argv[3][3:] or 10)
    idx = torch.arange(n, device="cuda") % m
    x, y = x[idx].contiguous(), y[idx].contiguous()
s = torch.cuda.current_stream().cuda_stream
def run():
    _lib.check(lib.sp_pedersen_batch_dev(x.data_ptr(), y.data_ptr(), o.data_ptr(), None, n, s), "ped")
for _ in range(2): run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5): run()
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 5
print("window_bits=%d table=%.1f GiB: %.3f ms per 2^%d hashes -> %.3e hashes/s" % (
    lib.sp_window_bits(), lib.sp_table_bytes() / 2**30, ms, logn, n / ms * 1e3))

#!/usr/bin/env python3
"""Kernel time of ONE level of n hashes (sp_pedersen_batch_dev), for the size classes of the latency kernels:
    python tools/quick_level.py [window_bits=26] n1 n2 ...      (HIP-event time per call, 20 calls each)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "stark-perpetual_amd"))
import torch
from starkperp import _lib
wb = int(sys.argv[1]) if len(sys.argv) > 1 else 26
sizes = [int(v) for v in sys.argv[2:]] or [20, 640, 1024, 1280, 2048, 2560, 4096]
lib = _lib.ensure_init(0, wb)
g = torch.Generator().manual_seed(1)
nmax = max(sizes)
def felts():
    t = torch.randint(-(2**63), 2**63 - 1, (nmax, 4), dtype=torch.int64, generator=g); t[:, 3] &= (1 << 58) - 1
    return t.cuda()
x, y = felts(), felts(); o = torch.empty_like(x)
s = torch.cuda.current_stream().cuda_stream
for n in sizes:
    run = lambda: _lib.check(lib.sp_pedersen_batch_dev(x.data_ptr(), y.data_ptr(), o.data_ptr(), None, n, s), "ped")
    for _ in range(3): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): run()
    e1.record(); torch.cuda.synchronize()
    print("n = %6d: %.2f us per level" % (n, e0.elapsed_time(e1) / 20 * 1e3))

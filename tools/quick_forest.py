#!/usr/bin/env python3
"""Forest build timing (dev aid)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "stark-perpetual_amd"))
import torch
from starkperp import _lib
lib = _lib.ensure_init(0)
st = torch.cuda.current_stream().cuda_stream
H = 16
for lt in (0, 2, 4, 5, 6):
    n0 = 1 << (H + lt)
    total = sum(n0 >> k for k in range(H + 1))
    g = torch.Generator().manual_seed(3)
    lv = torch.zeros((total, 4), dtype=torch.int64, device="cuda")
    t = torch.randint(-(2**63), 2**63 - 1, (n0, 4), dtype=torch.int64, generator=g); t[:, 3] &= (1 << 58) - 1
    lv[:n0] = t.cuda()
    def run(): _lib.check(lib.sp_merkle_forest_dev(lv.data_ptr(), 1 << lt, H, None, st), "forest")
    run(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): run()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    trees = 1 << lt
    print("forest of %d trees: %.3f ms -> %.3f ms/tree, %.3e hashes/s" % (trees, ms, ms / trees, trees * 65535 / ms * 1e3))

#!/usr/bin/env python3
"""Dense forests through sp_merkle_forest_dev (small levels four per launch, ped_top_kernel) against the same levels
hashed one by one with sp_pedersen_batch_dev (never fused): every node of every level, trees of every height 1 .. 13 and
forests of 1 .. 23 trees, under the window plan given by STARKPERP_WINDOW_BITS (dev aid / soak).
    STARKPERP_WINDOW_BITS=21 python tools/check_top_fusion.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "stark-perpetual_amd"))
import torch
from starkperp import _lib
lib = _lib.ensure_init(0, int(os.environ.get("STARKPERP_WINDOW_BITS", "26")))
st = torch.cuda.current_stream().cuda_stream
g = torch.Generator().manual_seed(5)
bad = checked = 0
for height in range(1, 14):
    for trees in (1, 2, 3, 5, 20, 23):
        n0 = trees << height
        total = trees * ((2 << height) - 1)
        lv = torch.zeros((total, 4), dtype=torch.int64, device="cuda")
        t = torch.randint(-(2**63), 2**63 - 1, (n0, 4), dtype=torch.int64, generator=g)
        t[:, 3] &= (1 << 58) - 1
        lv[:n0] = t.cuda()
        ref = lv.clone()
        _lib.check(lib.sp_merkle_forest_dev(lv.data_ptr(), trees, height, None, st), "forest")
        off, n = 0, n0
        for k in range(height):
            x = ref[off : off + n : 2].contiguous()
            y = ref[off + 1 : off + n : 2].contiguous()
            out = torch.zeros((n // 2, 4), dtype=torch.int64, device="cuda")
            _lib.check(lib.sp_pedersen_batch_dev(x.data_ptr(), y.data_ptr(), out.data_ptr(), None, n // 2, st), "batch")
            ref[off + n : off + n + n // 2] = out
            off += n
            n //= 2
        torch.cuda.synchronize()
        checked += total - n0
        if not bool((lv == ref).all()):
            bad += 1
            print("MISMATCH height", height, "trees", trees)
print("window bits %d: %d inner nodes of forests of 1 .. 23 trees, heights 1 .. 13, compared with unfused levels: %d mismatching forests"
      % (int(lib.sp_window_bits()), checked, bad))
sys.exit(1 if bad else 0)

#!/usr/bin/env python3
"""Full proof (commitments + Fiat-Shamir + query openings) timing (development aid)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "stark-perpetual_amd"))
import torch
from starkperp import stark
m = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
g = torch.Generator().manual_seed(3)
def felts(n):
    t = torch.randint(-(2**63), 2**63 - 1, (n, 4), dtype=torch.int64, generator=g); t[:, 3] &= (1 << 58) - 1
    return t.cuda()
xs, ys = felts(m), felts(m)
for it in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    proof = stark.prove(xs, ys, n_queries=8, seed=it)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("prove %d hashes (%d rows), 8 queries: %.1f ms" % (m, 512 * m, dt * 1e3))

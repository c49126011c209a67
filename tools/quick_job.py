#!/usr/bin/env python3
"""One 2^20-row AIR+FRI commit job, 3 times (run under rocprofv3 --kernel-trace --stats)."""
import os, sys, time, random
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "stark-perpetual_amd"))
import torch
from starkperp import _lib, stark
lib = _lib.ensure_init(0, 26)
m = 2048
g = torch.Generator().manual_seed(1)
def felts(n):
    t = torch.randint(-(2**63), 2**63 - 1, (n, 4), dtype=torch.int64, generator=g); t[:, 3] &= (1 << 58) - 1
    return t.cuda()
xs, ys = felts(m), felts(m)
rng = random.Random(2)
alphas = [rng.randrange(stark.FIELD_PRIME) for _ in range(stark.N_CONSTRAINTS)]
betas = [rng.randrange(stark.FIELD_PRIME) for _ in range(16)]
for _ in range(3):
    stark.prove_commitments(xs, ys, alphas, betas)
torch.cuda.synchronize()
t0 = time.perf_counter()
stark.prove_commitments(xs, ys, alphas, betas)
torch.cuda.synchronize()
print("job %.2f ms" % ((time.perf_counter() - t0) * 1e3))

#!/usr/bin/env python3
"""Staggered groups: group g+1's bulk levels start when group g's bulk levels are done, so the
latency-bound upper levels of one group run beside the bulk levels of the next (dev aid)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "stark-perpetual_amd"))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch
from starkperp import _lib
lib = _lib.ensure_init(0, 26)
H = 16

def forest(trees, seed):
    n0 = trees << H
    total = trees * ((2 << H) - 1)
    g = torch.Generator().manual_seed(seed)
    lv = torch.zeros((total, 4), dtype=torch.int64, device="cuda")
    t = torch.randint(-(2**63), 2**63 - 1, (n0, 4), dtype=torch.int64, generator=g)
    t[:, 3] &= (1 << 58) - 1
    lv[:n0] = t.cuda()
    return lv

def run(plan, nstreams, split, reps=5, check=False):
    streams = [torch.cuda.Stream() for _ in range(nstreams)]
    bufs = [forest(nb, 3 + i) for i, nb in enumerate(plan)]
    def go():
        prev = None
        for i, nb in enumerate(plan):
            s = streams[i % nstreams]
            with torch.cuda.stream(s):
                if prev is not None:
                    s.wait_event(prev)
                _lib.check(lib.sp_merkle_forest_dev(bufs[i].data_ptr(), nb << (H - split), split, None, s.cuda_stream), "forest")
                ev = torch.cuda.Event(); ev.record(s); prev = ev
                # upper part: the level-`split` array as the leaves of nb trees of height H - split
                off = sum((nb << H) >> k for k in range(split))
                _lib.check(lib.sp_merkle_forest_dev(bufs[i].data_ptr() + 32 * off, nb, H - split, None, s.cuda_stream), "forest")
    go(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        go()
        torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    print("plan %s streams %d split %d: %.3f ms  %.3e hashes/s" % (plan, nstreams, split, dt * 1e3, sum(plan) * 65535 / dt))
    return bufs

ref = forest(5, 3)
_lib.check(lib.sp_merkle_forest_dev(ref.data_ptr(), 5, H, None, torch.cuda.current_stream().cuda_stream), "f")
torch.cuda.synchronize()
b = run([5, 5, 5, 5], 4, 4)
print("roots equal:", bool((b[0][-5:] == ref[-5:]).all()))
run([20], 1, 4)
run([10, 10], 2, 4)
run([10, 10], 2, 3)
run([5, 5, 5, 5], 4, 3)
run([5, 5, 5, 5], 4, 5)
run([4, 4, 4, 4, 4], 5, 4)
run([7, 7, 6], 3, 4)
run([2] * 10, 5, 3)
run([2] * 10, 8, 2)

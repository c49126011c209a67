#!/usr/bin/env python3
"""Do two lockstep forests on two HIP streams overlap when one stream has priority? (dev aid)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "stark-perpetual_amd"))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch
from starkperp import _lib
lib = _lib.ensure_init(0, 26)
H = 16
try:
    print("priority range", torch.cuda.Stream.priority_range())
except Exception as e:
    print("no priority_range", e)

def forest(trees, seed):
    n0 = trees << H
    total = trees * ((2 << H) - 1)
    g = torch.Generator().manual_seed(seed)
    lv = torch.zeros((total, 4), dtype=torch.int64, device="cuda")
    t = torch.randint(-(2**63), 2**63 - 1, (n0, 4), dtype=torch.int64, generator=g)
    t[:, 3] &= (1 << 58) - 1
    lv[:n0] = t.cuda()
    return lv

def run(plan, prios, reps=5):
    streams = [torch.cuda.Stream(priority=p) for p in prios]
    bufs = [forest(nb, 3 + i) for i, nb in enumerate(plan)]
    def go():
        for i, nb in enumerate(plan):
            s = streams[i % len(streams)]
            _lib.check(lib.sp_merkle_forest_dev(bufs[i].data_ptr(), nb, H, None, s.cuda_stream), "forest")
    go(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        go()
        torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    print("plan %s prios %s: %.3f ms  %.3e hashes/s" % (plan, prios, dt * 1e3, sum(plan) * 65535 / dt))

run([20], [0])
run([10, 10], [0, 0])
run([10, 10], [-1, 0])
run([5, 5, 5, 5], [-1, 0])
run([5, 5, 5, 5], [-1, 0, 0, 0])
run([7, 7, 6], [-1, 0, 0])
run([4, 16], [-1, 0])
run([16, 4], [0, -1])

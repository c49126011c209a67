#!/usr/bin/env python3
"""Do a small forest's whole rebuild and a large forest's latency-bound top overlap on two streams? (dev aid)

  python tools/quick_overlap.py            # plans below, 20 trees in total

Measured (round 2, profiles/r02_overlap_experiment.txt): no plan beats the lockstep forest of 20 (2.20 ms); the best
split, [12, 8] on two streams, takes 2.24 ms.  The same with `s_setprio 3` in the latency kernels and the bulk launches
capped at two waves per SIMD by 80 KB of dynamic LDS (so that a latency wave always finds registers): no change -
both knobs were removed again."""
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


def bulk_and_top(buf, nb, split, stream):
    """levels 0..split-1 as one forest of height `split`, then the rest; returns the event after the bulk part"""
    _lib.check(lib.sp_merkle_forest_dev(buf.data_ptr(), nb << (H - split), split, None, stream.cuda_stream), "forest")
    ev = torch.cuda.Event()
    ev.record(stream)
    off = sum((nb << H) >> k for k in range(split))
    _lib.check(lib.sp_merkle_forest_dev(buf.data_ptr() + 32 * off, nb, H - split, None, stream.cuda_stream), "forest")
    return ev


def run(name, plan, mode, reps=6):
    """mode 'seq': one stream, one after the other.  'after_bulk': group g + 1 starts (on its own stream) when the
    bulk levels of group g are done.  'together': every group on its own stream from the start."""
    streams = [torch.cuda.Stream(priority=(-1 if (mode != "seq" and i > 0) else 0)) for i in range(len(plan))]
    bufs = [forest(nb, 3 + i) for i, nb in enumerate(plan)]

    def go():
        prev = None
        for i, nb in enumerate(plan):
            s = streams[0] if mode == "seq" else streams[i]
            with torch.cuda.stream(s):
                if mode == "after_bulk" and prev is not None:
                    s.wait_event(prev)
                prev = bulk_and_top(bufs[i], nb, 4, s)

    go()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter()
        go()
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    print("%-34s %-10s %.3f ms  %.3e hashes/s" % (name + " " + str(plan), mode, best * 1e3, sum(plan) * 65535 / best))
    return bufs


ref = forest(4, 4)
_lib.check(lib.sp_merkle_forest_dev(ref.data_ptr(), 4, H, None, torch.cuda.current_stream().cuda_stream), "f")
torch.cuda.synchronize()
b = run("check", [16, 4], "after_bulk", reps=2)
print("roots of the 4-tree group equal the lockstep ones:", bool((b[1][-4:] == ref[-4:]).all()))
run("lockstep", [20], "seq")
run("two calls", [16, 4], "seq")
for plan in ([16, 4], [4, 16], [16, 2, 2], [8, 8, 4], [16, 3, 1], [12, 8], [18, 2]):
    for mode in ("after_bulk", "together"):
        run("", plan, mode)

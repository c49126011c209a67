#!/usr/bin/env python3
"""Co-scheduling the latency-bound top of one group of trees beside the bulk levels of the next on DISJOINT compute
units: streams created with hipExtStreamCreateWithCUMask (dev aid, verdict round 2 item 1a).

Streams and priorities cannot do this (profiles/r02_overlap_experiment.txt): the bulk kernel fills every wave slot,
a latency wave waits ~190 us for one.  A CU mask reserves `R` compute units per XCD for the tail stream; the bulk
stream gets the rest.  Bit i of a mask is CU i / 8 of XCD i % 8 (the KFD interleaves the user mask over the XCCs).

  python tools/quick_cumask.py                       # 20 trees of 2^16 leaves, plans below
  STARKPERP_SPLIT_LANES=63488 STARKPERP_FINISH_LANES=63488 python tools/quick_cumask.py   # rounds of 248 CUs

Every plan's roots are compared with the lockstep forest's."""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "stark-perpetual_amd"))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch
from starkperp import _lib
lib = _lib.ensure_init(0, 26)
hip = ctypes.CDLL("libamdhip64.so")
H = 16
N_CU = torch.cuda.get_device_properties(0).multi_processor_count
N_XCD = 8


def masked_stream(cus):
    """A stream confined to the compute units in `cus` (mask bit indices)."""
    words = (N_CU + 31) // 32
    m = (ctypes.c_uint32 * words)()
    for c in cus:
        m[c // 32] |= 1 << (c % 32)
    s = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(s), ctypes.c_uint32(words), m)
    assert rc == 0, "hipExtStreamCreateWithCUMask rc=%d" % rc
    return s


def plain_stream():
    s = ctypes.c_void_p()
    assert hip.hipStreamCreateWithFlags(ctypes.byref(s), ctypes.c_uint(1)) == 0  # hipStreamNonBlocking
    return s


def event():
    e = ctypes.c_void_p()
    assert hip.hipEventCreateWithFlags(ctypes.byref(e), ctypes.c_uint(2)) == 0  # hipEventDisableTiming
    return e


def forest(trees, seed):
    n0 = trees << H
    total = trees * ((2 << H) - 1)
    g = torch.Generator().manual_seed(seed)
    lv = torch.zeros((total, 4), dtype=torch.int64, device="cuda")
    t = torch.randint(-(2**63), 2**63 - 1, (n0, 4), dtype=torch.int64, generator=g)
    t[:, 3] &= (1 << 58) - 1
    lv[:n0] = t.cuda()
    return lv


def levels(buf, nb, lo, hi, stream):
    """levels lo..hi-1 of a forest of nb trees (level `lo`'s inputs are in place): a forest of nb << (H - hi)
    trees of height hi - lo over the level-lo array"""
    off = sum((nb << H) >> k for k in range(lo))
    _lib.check(lib.sp_merkle_forest_dev(buf.data_ptr() + 32 * off, nb << (H - hi), hi - lo, None, stream), "forest")


def roots(buf, nb):
    return buf[-nb:].clone()


def timed(go, reps=8):
    go()
    torch.cuda.synchronize()
    best, all_t = 1e9, []
    for _ in range(reps):
        t0 = time.perf_counter()
        go()
        torch.cuda.synchronize()
        all_t.append(time.perf_counter() - t0)
    all_t.sort()
    return all_t[0], all_t[len(all_t) // 2]


def report(name, t, n_trees=20):
    print("%-78s best %.3f ms  median %.3f ms  %.3e hashes/s (best)" % (name, t[0] * 1e3, t[1] * 1e3, n_trees * 65535 / t[0]),
          flush=True)


# reference: lockstep forest of 20 on an ordinary stream
s_plain = plain_stream()
bufs20 = forest(20, 3)
report("lockstep [20], ordinary stream", timed(lambda: levels(bufs20, 20, 0, H, s_plain)))
ref_roots = roots(bufs20, 20)
halves = [forest(10, 3), forest(10, 3)]
# the same leaves as the 20-tree forest, ten trees each
halves[0][: 10 << H] = bufs20[: 10 << H]
halves[1][: 10 << H] = bufs20[10 << H: 20 << H]


def check(name):
    got = torch.cat([roots(halves[0], 10), roots(halves[1], 10)])
    ok = bool((got == ref_roots).all())
    if not ok:
        print("   ROOTS DIFFER in", name)
    return ok


def split_plan(s_bulk, s_tail, cut, tail_both=False):
    """group 0: levels [0, cut) on s_bulk, [cut, H) on s_tail; group 1: everything on s_bulk behind group 0's
    bulk part (stream order).  tail_both: group 1's top on s_tail as well."""
    ev = event()
    ev2 = event()

    def go():
        levels(halves[0], 10, 0, cut, s_bulk)
        hip.hipEventRecord(ev, s_bulk)
        hip.hipStreamWaitEvent(s_tail, ev, 0)
        levels(halves[0], 10, cut, H, s_tail)
        if tail_both:
            levels(halves[1], 10, 0, cut, s_bulk)
            hip.hipEventRecord(ev2, s_bulk)
            hip.hipStreamWaitEvent(s_tail, ev2, 0)
            levels(halves[1], 10, cut, H, s_tail)
        else:
            levels(halves[1], 10, 0, H, s_bulk)
    return go


# two halves one after the other on one stream (what the split alone costs)
report("[10, 10] one ordinary stream, sequential",
       timed(lambda: (levels(halves[0], 10, 0, H, s_plain), levels(halves[1], 10, 0, H, s_plain))))
check("sequential halves")
s_plain2 = plain_stream()
for cut in (8, 6):
    name = "[10, 10] two ordinary streams, top of group 0 from level %d beside group 1" % cut
    report(name, timed(split_plan(s_plain, s_plain2, cut)))
    check(name)
for R in (1, 2, 4):
    tail_cus = [x + N_XCD * k for k in range(R) for x in range(N_XCD)]
    bulk_cus = [c for c in range(N_CU) if c not in tail_cus]
    s_tail = masked_stream(tail_cus)
    s_bulk = masked_stream(bulk_cus)
    report("lockstep [20] on the bulk stream alone (%d of %d CUs)" % (len(bulk_cus), N_CU),
           timed(lambda: levels(bufs20, 20, 0, H, s_bulk)))
    for cut in (8, 6):
        name = "[10, 10] CU masks %d + %d, top of group 0 from level %d on the tail CUs" % (len(bulk_cus), len(tail_cus), cut)
        report(name, timed(split_plan(s_bulk, s_tail, cut)))
        check(name)
    # group 1's bulk on the masked stream only while group 0's top runs: everything else on all CUs
    ev_a, ev_b = event(), event()

    def go3(cut=8):
        levels(halves[0], 10, 0, cut, s_plain)
        hip.hipEventRecord(ev_a, s_plain)
        hip.hipStreamWaitEvent(s_tail, ev_a, 0)
        hip.hipStreamWaitEvent(s_bulk, ev_a, 0)
        levels(halves[0], 10, cut, H, s_tail)
        levels(halves[1], 10, 0, 2, s_bulk)          # the two largest levels beside group 0's top
        hip.hipEventRecord(ev_b, s_bulk)
        hip.hipStreamWaitEvent(s_plain, ev_b, 0)
        levels(halves[1], 10, 2, H, s_plain)
    name = "[10, 10] all CUs except: group 1 levels 0-1 on %d CUs beside group 0's top (from 8) on %d" % (len(bulk_cus), len(tail_cus))
    report(name, timed(go3))
    check(name)

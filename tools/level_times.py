#!/usr/bin/env python3
"""Per-level launch durations of one lockstep forest rebuild (dev aid).

  python tools/level_times.py run <trees> [window_bits]    # builds the forest 3 times (run under rocprofv3 --kernel-trace)
  python tools/level_times.py parse <kernel_trace.csv> <launches_per_build> [all]   # all: every kernel, not only ped_*
"""
import csv
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "stark-perpetual_amd"))


def run(trees, wbits):
    import torch
    from starkperp import _lib
    lib = _lib.ensure_init(0, wbits or None)
    st = torch.cuda.current_stream().cuda_stream
    H = 16
    n0 = trees << H
    total = trees * ((2 << H) - 1)
    g = torch.Generator().manual_seed(3)
    lv = torch.zeros((total, 4), dtype=torch.int64, device="cuda")
    t = torch.randint(-(2**63), 2**63 - 1, (n0, 4), dtype=torch.int64, generator=g)
    t[:, 3] &= (1 << 58) - 1
    lv[:n0] = t.cuda()
    for _ in range(3):
        _lib.check(lib.sp_merkle_forest_dev(lv.data_ptr(), trees, H, None, st), "forest")
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        _lib.check(lib.sp_merkle_forest_dev(lv.data_ptr(), trees, H, None, st), "forest")
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    print("forest of %d trees: %.3f ms per build, %.3e hashes/s" % (trees, ms, trees * 65535 / ms * 1e3))
    if os.environ.get("LEVEL_TIMES_SUSTAINED"):  # 400 builds back to back, then 400 timed: the steady state
        for rep in range(2):
            e0.record()
            for _ in range(400):
                _lib.check(lib.sp_merkle_forest_dev(lv.data_ptr(), trees, H, None, st), "forest")
            e1.record()
            torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 400
        print("sustained: %.4f ms per build, %.4e hashes/s" % (ms, trees * 65535 / ms * 1e3))
    if os.environ.get("LEVEL_TIMES_GRAPH"):
        # the launches of one forest build captured into a hipGraph and replayed - what the
        # launch boundaries cost when the host is out of the picture
        side = torch.cuda.Stream()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.stream(side):
            _lib.check(lib.sp_merkle_forest_dev(lv.data_ptr(), trees, H, None, side.cuda_stream), "forest")  # scratch of this stream
            side.synchronize()
            graph.capture_begin()
            _lib.check(lib.sp_merkle_forest_dev(lv.data_ptr(), trees, H, None, side.cuda_stream), "forest")
            graph.capture_end()
        torch.cuda.synchronize()
        for rep in range(2):
            e0.record()
            for _ in range(400):
                graph.replay()
            e1.record()
            torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 400
        print("hipGraph replay: %.4f ms per build, %.4e hashes/s" % (ms, trees * 65535 / ms * 1e3))


def parse(path, tail, every_kernel=False):
    rows = [r for r in csv.DictReader(open(path)) if every_kernel or "ped_" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    rows = rows[-tail:]
    t0 = int(rows[0]["Start_Timestamp"])
    prev_end = t0
    for r in rows:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        name = r["Kernel_Name"].split("(")[0].replace("void sp::", "").replace("sp::", "")
        print("%9.1f us  gap %6.1f  dur %7.1f  grid %8s  %s" % ((s - t0) / 1e3, (s - prev_end) / 1e3, (e - s) / 1e3,
                                                            r.get("Grid_Size", r.get("Grid_Size_X", "?")), name))
        prev_end = e
    print("total %.1f us" % ((prev_end - t0) / 1e3))


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run(int(sys.argv[2]), int(sys.argv[3]) if len(sys.argv) > 3 else 0)
    else:
        parse(sys.argv[2], int(sys.argv[3]), len(sys.argv) > 4 and sys.argv[4] == "all")

#!/usr/bin/env python3
"""Average duration of the accumulate-kernel launches of bench.py's TIMED region in a rocprofv3
kernel trace, to compare with roofline.avg_launch_us of the same run.

  python tools/trace_timed_avg.py <kernel_trace.csv> <n> [grid,grid,...]

n = roofline.launches of the traced run.  The timed region's pure ped_accumulate_kernel launches are levels 0 and 1
of every 20-tree forest (grids 655360 and 327680 lanes); since round 6 the driver's command is traced WITH its later
legs (the AIR + FRI jobs and the bulk batch launch the same kernel at other sizes), so the launches are selected by
grid size first and the last n of those are the sustained window (the burst and the pre-heat come before it)."""
import csv
import sys


def main():
    path, n = sys.argv[1], int(sys.argv[2])
    grids = set(int(g) for g in (sys.argv[3] if len(sys.argv) > 3 else "655360,327680").split(","))
    rows = [r for r in csv.DictReader(open(path)) if "ped_accumulate_kernel(" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    sel = [r for r in rows if int(r.get("Grid_Size", r.get("Grid_Size_X", 0)) or 0) in grids]
    us = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in sel]
    print("ped_accumulate_kernel launches in trace: %d; of the forest's level-0 / level-1 sizes %s: %d, average %.1f us" % (
        len(rows), sorted(grids), len(us), sum(us) / max(len(us), 1)))
    print("last %d of those (the sustained window of the timed regions): average %.1f us" % (n, sum(us[-n:]) / max(len(us[-n:]), 1)))


if __name__ == "__main__":
    main()

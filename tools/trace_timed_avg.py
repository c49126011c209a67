#!/usr/bin/env python3
"""Average duration of the accumulate-kernel launches of bench.py's TIMED region in a rocprofv3
kernel trace (the last `n` ped_accumulate* launches of the process; bench.py reports n as
roofline.launches), to compare with roofline.avg_launch_us of the same run.

  python tools/trace_timed_avg.py gpurun_out/r01i_stats/b_kernel_trace.csv 32
"""
import csv
import sys


def main():
    path, n = sys.argv[1], int(sys.argv[2])
    rows = [r for r in csv.DictReader(open(path)) if "ped_accumulate_kernel(" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    us = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows]
    print("accumulate launches in trace: %d, average %.1f us" % (len(us), sum(us) / len(us)))
    print("last %d (the timed region): average %.1f us" % (n, sum(us[-n:]) / n))


if __name__ == "__main__":
    main()

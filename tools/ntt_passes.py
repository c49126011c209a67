#!/usr/bin/env python3
"""Per-launch durations of ntt_tile_kernel in a 4-column LDE 2^20 -> 2^22 from a rocprofv3 kernel trace of
tools/quick_lde.py (6 LDEs: 1 warm-up + 5 timed; 5 launches each: 2 inverse passes over 2^20 x 4, 3 forward passes
over 2^22 x 4):   python tools/ntt_passes.py <kernel_trace.csv> [label] [launches per LDE]"""
import csv
import sys


def main():
    rows = []
    for r in csv.DictReader(open(sys.argv[1])):
        if "ntt_tile_kernel" in r["Kernel_Name"]:
            rows.append((int(r["Start_Timestamp"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
    rows.sort()
    per = int(sys.argv[3]) if len(sys.argv) > 3 else 5  # launches per LDE (6 with 1024-felt tiles: 3 + 3 passes)
    ldes = [rows[i:i + per] for i in range(0, len(rows) - per + 1, per)][1:]  # drop the warm-up LDE
    label = sys.argv[2] if len(sys.argv) > 2 else ""
    if not ldes:
        print("%s: no ntt_tile_kernel launches" % label)
        return
    avg = [sum(l[j][1] for l in ldes) / len(ldes) for j in range(per)]
    span = sum((l[-1][0] + l[-1][1] * 1e3 - l[0][0]) / 1e3 for l in ldes) / len(ldes)
    small, big, first = 2 * 4 * (1 << 20) * 32, 2 * 4 * (1 << 22) * 32, 4 * ((1 << 20) + (1 << 22)) * 32
    gb = [small] * (per - 3) + [first, big, big]
    print("%-34s launches (us): %s   sum %.1f   first-start..last-end %.1f" % (
        label, " ".join("%7.1f" % a for a in avg), sum(avg), span))
    print("%-34s algorithmic TB/s: %s" % ("", " ".join("%7.2f" % (b / (a * 1e-6) / 1e12) for a, b in zip(avg, gb))))


if __name__ == "__main__":
    main()

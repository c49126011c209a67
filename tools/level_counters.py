#!/usr/bin/env python3
"""Per-launch SQ counters of one lockstep forest rebuild, joined with the launch durations (dev aid).

  rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES --kernel-trace --output-format csv \
      -d gpurun_out/lc -o t -- python tools/level_times.py run <trees> 26
  python tools/level_counters.py gpurun_out/lc/t_counter_collection.csv <launches_per_build>

Prints, per launch of the LAST build: grid, duration, waves, VALU instructions per wave (the dependent
chain a level costs when it cannot fill the chip) and ns per instruction of that chain."""
import collections
import csv
import sys


def main():
    path, tail = sys.argv[1], int(sys.argv[2])
    rows = collections.OrderedDict()
    for r in csv.DictReader(open(path)):
        if "ped_" not in r["Kernel_Name"] and "tree_" not in r["Kernel_Name"]:
            continue
        d = rows.setdefault(int(r["Dispatch_Id"]), {"name": r["Kernel_Name"].split("(")[0].replace("void sp::", "").replace("sp::", ""),
                                                    "grid": int(r["Grid_Size"]), "vgpr": r.get("VGPR_Count", "?"),
                                                    "start": int(r["Start_Timestamp"]), "end": int(r["End_Timestamp"])})
        d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    items = sorted(rows.values(), key=lambda d: d["start"])[-tail:]
    print("%-44s %9s %5s %8s %8s %10s %8s" % ("kernel", "grid", "vgpr", "dur_us", "waves", "valu/wave", "ns/instr"))
    for d in items:
        dur = (d["end"] - d["start"]) / 1e3
        waves = d.get("SQ_WAVES", 0.0)
        ipw = d.get("SQ_INSTS_VALU", 0.0) / waves if waves else 0.0
        per_simd = max(1.0, waves / 1024.0)
        print("%-44s %9d %5s %8.1f %8d %10.0f %8.2f" % (d["name"][:44], d["grid"], d["vgpr"], dur, waves, ipw,
                                                      dur * 1e3 / (ipw * per_simd) if ipw else 0.0))


if __name__ == "__main__":
    main()

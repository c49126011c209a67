#!/usr/bin/env python3
"""SQ_INSTS_VALU per hash of the bulk hash kernels from a rocprofv3 PMC pass of tools/bulk_only.py:

  rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/sq -o b \
      -- python tools/bulk_only.py 22 26
  python tools/valu_counts.py gpurun_out/sq/b_counter_collection.csv gpurun_out/sq/b_kernel_trace.csv 22 26

Prints one JSON object (window_bits entry of profiles/r0N_valu_issue.json).  SQ_INSTS_VALU is wave-level:
one hash per lane in ped_accumulate_kernel, so instructions per hash = counter / waves."""
import collections
import csv
import json
import sys


def main():
    counters, trace, logn, wbits = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
    n = 1 << logn
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(counters)):
        name = r["Kernel_Name"].split("(")[0].replace("void ", "")
        acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
    dur = collections.defaultdict(list)
    for r in csv.DictReader(open(trace)):
        name = r["Kernel_Name"].split("(")[0].replace("void ", "")
        dur[name].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
    out = {"window_bits": wbits, "hashes_per_launch": n}
    for kernel, key in (("sp::ped_accumulate_kernel", "accumulate"), ("sp::ped_finish_kernel", "finish")):
        v = acc[kernel]["SQ_INSTS_VALU"]
        if not v:
            continue
        per_launch = sum(v[-3:]) / len(v[-3:])
        out[key + "_instr_per_hash"] = round(per_launch / n * 64 / 64 * 1.0, 1) if key == "finish" else round(per_launch / (n / 64), 1)
        if key == "finish":  # finish threads own several hashes: per hash = wave instructions * 64 lanes / hashes ... reported per hash-lane equivalent
            out[key + "_instr_per_hash"] = round(per_launch * 64 / n, 1)
        out[key + "_kernel_ms_under_pmc"] = round(sum(dur[kernel][-3:]) / len(dur[kernel][-3:]), 3)
        g = acc[kernel].get("GRBM_GUI_ACTIVE")
        if g and key == "accumulate":
            out["grbm_gui_active_cycles"] = g[-1]
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()

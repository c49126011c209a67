#!/usr/bin/env python3
"""Per-item VALU instruction counts of the kernels configs[2] is made of, from a rocprofv3 PMC pass of
`tools/c3_probe.py pmc` (SQ_INSTS_VALU SQ_WAVES) plus the kernel trace of the same pass:

    python tools/c3_counters.py <counter_collection.csv> <kernel_trace.csv> [out.json]

SQ_INSTS_VALU counts wave-level instructions.  Verification kernels: one signature per lane, so instructions per
signature = SQ_INSTS_VALU / SQ_WAVES (every lane of a wave executes the wave's instruction stream).  ped_chain /
ped_path: a lane GROUP of 4 x 2^LOG_Q lanes (x 2^dup copies) computes one hash per step; the parser reports
instructions per wave and per launch, the item counts come from the probe (4096 chains x 3 steps; paths as launched)."""
import collections
import csv
import json
import sys


def main():
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(sys.argv[1])):
        name = r["Kernel_Name"].split("(")[0].replace("void ", "")
        acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
    dur = collections.defaultdict(list)
    grid = collections.defaultdict(list)
    for r in csv.DictReader(open(sys.argv[2])):
        name = r["Kernel_Name"].split("(")[0].replace("void ", "")
        dur[name].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
        grid[name].append(int(r.get("Grid_Size", r.get("Grid_Size_X", 0)) or 0))
    out = {"_source": "rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES --kernel-trace -- python tools/c3_probe.py pmc", "kernels": {}}
    for name in sorted(acc):
        if not any(t in name for t in ("ecdsa_verify", "ped_chain", "ped_path", "ped_quad", "tree_")):
            continue
        v, w = acc[name].get("SQ_INSTS_VALU", []), acc[name].get("SQ_WAVES", [])
        if not v or not w:
            continue
        k = {"launches": len(v), "valu_wave_instr_per_launch_last": v[-1], "waves_per_launch_last": w[-1],
             "instr_per_wave_last": v[-1] / max(w[-1], 1), "waves_per_simd_last": w[-1] / 1024.0,
             "duration_us_under_pmc_last": dur[name][-1] if dur[name] else None,
             "grid_last": grid[name][-1] if grid[name] else None}
        if "ecdsa_verify" in name:
            k["instr_per_signature"] = k["instr_per_wave_last"]
        out["kernels"][name] = k
    text = json.dumps(out, indent=1)
    if len(sys.argv) > 3:
        open(sys.argv[3], "w").write(text + "\n")
    print(text)


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Per-item VALU instruction counts of the kernels configs[2] is made of, from a rocprofv3 PMC pass of
`tools/c3_probe.py pmc` (SQ_INSTS_VALU SQ_WAVES):

    python tools/c3_counters.py <counter_collection.csv> [out.json]

The counter file carries, per dispatch, the grid size, the register counts and the start / end timestamps, so one file
is enough.  SQ_INSTS_VALU counts wave-level instructions.  Verification kernels: one signature per lane, so instructions
per signature = SQ_INSTS_VALU / SQ_WAVES (every lane of a wave executes the wave's instruction stream).  ped_chain /
ped_path / ped_quad: a lane GROUP of 4 x 2^LOG_Q lanes computes one hash per step; reported per wave and per launch -
the item counts come from the probe (4096 chains x 3 steps; 4096 paths x the levels that do not merge).
Launches are grouped by grid size; the LAST launch of a size is the one reported (warm)."""
import collections
import csv
import json
import sys


def main():
    disp = collections.OrderedDict()
    for r in csv.DictReader(open(sys.argv[1])):
        d = disp.setdefault(int(r["Dispatch_Id"]), {
            "name": r["Kernel_Name"].split("(")[0].replace("void ", ""), "grid": int(r["Grid_Size"]),
            "vgpr": int(r.get("VGPR_Count", 0) or 0), "agpr": int(r.get("Accum_VGPR_Count", 0) or 0),
            "scratch": int(r.get("Scratch_Size", 0) or 0),
            "us": (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3})
        d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    out = {"_source": "rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES --kernel-trace -- python tools/c3_probe.py pmc",
           "workload": "2 x ladder verification of 2^16 signatures, 2 x keyed verification of 2^16, 3 x sp_order_batch of 4096 "
                       "orders (keyed verification of 4096, ped_chain_kernel: 4096 chains x 3 hashes, ped_path_kernel: 4096 "
                       "paths through the levels that do not merge, ped_quad_kernel: the merging levels above)",
           "kernels": {}}
    for did in sorted(disp):
        d = disp[did]
        name = d["name"]
        if not any(t in name for t in ("ecdsa_verify", "ped_chain", "ped_path", "ped_quad", "tree_")):
            continue
        v, w = d.get("SQ_INSTS_VALU"), d.get("SQ_WAVES")
        if not v or not w:
            continue
        k = out["kernels"].setdefault(name, {"vgpr": d["vgpr"], "agpr": d["agpr"], "scratch_bytes_per_lane": d["scratch"],
                                             "by_grid": {}})
        e = k["by_grid"].setdefault(str(d["grid"]), {"launches": 0})
        e["launches"] += 1
        e.update({"grid": d["grid"], "waves": w, "waves_per_simd": w / 1024.0, "valu_wave_instr": v,
                  "instr_per_wave": v / w, "duration_us_under_pmc": d["us"]})
        if "ecdsa_verify" in name:
            e["instr_per_signature"] = v / w  # one signature per lane
    text = json.dumps(out, indent=1)
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(text + "\n")
    print(text)


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""profiles/r06_valu_issue.json: the per-hash SQ_INSTS_VALU counts of THIS round's binary on top of the issue-interval
model of round 4 (the opcode mix of ped_accumulate_kernel and the per-opcode intervals did not change: csrc/pedersen.hip's
accumulate body is the one profiles/r04_valu_issue.json priced).

    python tools/make_valu_issue.py <counts_w26.json> <counts_w21.json> <level_counters.txt> <evidence.json> > r06_valu_issue.json

counts_*.json: tools/valu_counts.py on a `--pmc SQ_INSTS_VALU SQ_WAVES GRBM_GUI_ACTIVE` pass of tools/bulk_only.py 22 <w>;
level_counters.txt: tools/level_counters.py on the 20-tree forest (the finish-lds launches the bulk batch never takes)."""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    base = json.load(open(os.path.join(ROOT, "profiles", "r04_valu_issue.json")))
    w26, w21 = json.load(open(sys.argv[1])), json.load(open(sys.argv[2]))
    ev = json.load(open(sys.argv[4]))
    out = dict(base)
    out["_source"] = ("round 6, tools/run_r06_prof.sh on the end-of-round binary (lib sha256 %s..., csrc sha256 %s...).  Counts: "
                      "rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES GRBM_GUI_ACTIVE --kernel-trace -- python tools/bulk_only.py 22 26 "
                      "(and 22 21), tools/valu_counts.py.  Issue interval (cycles_per_wave64_valu_instr, mix): carried over from "
                      "profiles/r04_valu_issue.json - the accumulate kernel's opcode mix and the per-opcode intervals "
                      "(profiles/r04_valu_rate_ubench.txt) are unchanged.  27-bit entry: carried over, not re-measured."
                      % (ev["lib_sha256"][:16], ev["csrc_sha256"][:16]))
    for w, m in (("26", w26), ("21", w21)):
        e = dict(base["window_bits"].get(w, {}))
        e["accumulate_instr_per_hash"] = m["accumulate_instr_per_hash"]
        if "finish_instr_per_hash" in m:
            e["finish_instr_per_hash"] = m["finish_instr_per_hash"]
        e["measured_round"] = 6
        for k in ("accumulate_kernel_ms_under_pmc", "finish_kernel_ms_under_pmc", "grbm_gui_active_cycles"):
            if k in m:
                e[k] = m[k]
        out["window_bits"][w] = e
    # the 20-tree forest launch by launch (tools/level_counters.py): wave-level VALU instructions of EVERY kernel of one
    # build / its 20 x 65535 hashes = what a hash of the timed region costs, finish-lds and latency-bound levels included
    launches, total = [], 0.0
    for line in open(sys.argv[3]):
        f = line.split()
        if len(f) >= 7 and f[0].startswith("ped_") and f[-6].isdigit():
            # kernel names may contain ", " (template arguments): the numeric columns are the last six
            name = " ".join(f[:-6])
            grid, dur, waves, ipw = int(f[-6]), float(f[-4]), int(f[-3]), float(f[-2])
            launches.append({"kernel": name, "grid": grid, "dur_us": dur, "waves": waves, "valu_per_wave": ipw})
            total += waves * ipw
    hashes = 20 * 65535
    out["forest_20"] = {"launches": launches, "valu_wave_instr_per_build": total, "hashes_per_build": hashes,
                        "instr_per_hash_all_kernels": total * 64.0 / hashes,
                        "note": "wave-level SQ_INSTS_VALU x 64 / hashes: the convention of accumulate_instr_per_hash"}
    acc = [l for l in launches if l["kernel"].startswith("ped_accumulate_kernel")]
    if acc:
        out["forest_20"]["accumulate_kernel_instr_per_hash"] = sum(l["waves"] * l["valu_per_wave"] for l in acc) / sum(l["waves"] for l in acc)
    json.dump(out, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()

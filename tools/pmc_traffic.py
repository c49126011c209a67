#!/usr/bin/env python3
"""Summarises two rocprofv3 PMC passes of bench.py (one with FETCH_SIZE, one with WRITE_SIZE) into
profiles/<out>.json (default r02_pmc_traffic.json) together with the configuration they ran:

  python tools/pmc_traffic.py <fetch.csv> <write.csv> [out_name] [config_key] [config text]

bench.py prints `traffic` from that file and says whether its own run has the same config_key.
Run on the GPU box:

  cd /tmp && export TMPDIR=/tmp
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_fetch -o b -- python bench.py --no-cpu-baseline --no-extras
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_write -o b -- python bench.py --no-cpu-baseline --no-extras
  python tools/pmc_traffic.py gpurun_out/pmc_fetch/b_counter_collection.csv gpurun_out/pmc_write/b_counter_collection.csv

Units (MI355X_MICROARCH.md, HBM section): FETCH_SIZE / WRITE_SIZE are in KiB.  The guide's gfx950
note (FETCH_SIZE reads half the bytes of a WIDE COALESCED stream) does not apply to these kernels,
whose reads are 64-byte random table gathers: calibrated on the known byte count of level 0 of the
bench tree (32768 hashes x (24 gathers x 64 B + 64 B of inputs) = 52.4 MB; raw FETCH_SIZE 53.5 MB),
so FETCH_SIZE is used as reported.  WRITE_SIZE calibrated exact on torch's fill kernel (4.19 MB).
"""
import collections
import csv
import json
import sys


def per_kernel(path, counter):
    """kernel -> list of counter values in dispatch order"""
    rows = [r for r in csv.DictReader(open(path)) if r["Counter_Name"] == counter]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    acc = collections.defaultdict(list)
    for r in rows:
        acc[r["Kernel_Name"].split("(")[0].replace("void ", "")].append(float(r["Counter_Value"]))
    return acc


import os

# bulk launches of the timed regions of the profiled `bench.py --steps 20` run: round 2 timed ONE forest (levels
# 0..3 = 4 launches); round 3 repeats the region until 50 ms are timed and only levels 0 and 1 are pure
# ped_accumulate_kernel launches (2 per region) - pass the count bench.py printed as roofline.launches
TIMED_LAUNCHES = int(os.environ.get("PMC_TIMED_LAUNCHES", "4"))


def main():
    fetch = per_kernel(sys.argv[1], "FETCH_SIZE")
    write = per_kernel(sys.argv[2], "WRITE_SIZE")
    out = {"units": "bytes; FETCH_SIZE KiB x 1024 (calibrated on the 64-B gathers, see docstring), WRITE_SIZE KiB x 1024",
           "kernels": {}}
    tot_b, tot_n = 0.0, 0
    for name in sorted(set(fetch) | set(write)):
        f, w = fetch.get(name, []), write.get(name, [])
        n = max(len(f), len(w), 1)
        b = (sum(f) * 1024 + sum(w) * 1024) / n
        out["kernels"][name] = {"launches": n, "fetch_bytes_per_launch": sum(f) * 1024 / n,
                                "write_bytes_per_launch": sum(w) * 1024 / n, "hbm_bytes_per_launch": b}
        if any(t in name for t in ("ntt_tile_kernel", "air_eval", "fri_fold_kernel")):
            # wide coalesced streams: FETCH_SIZE reports half the bytes on gfx950 (MI355X_MICROARCH.md, HBM)
            out["kernels"][name]["hbm_bytes_per_launch_fetch_doubled"] = (2 * sum(f) + sum(w)) * 1024 / n
            out["kernels"][name]["note"] = "16-byte-per-lane coalesced reads: FETCH_SIZE doubled per the gfx950 note"
        merkle_run = len(sys.argv) > 4 and sys.argv[4].startswith("merkle")
        if merkle_run and name == "sp::ped_accumulate_kernel" and len(f) >= TIMED_LAUNCHES and len(w) >= TIMED_LAUNCHES:
            # the launches bench.py times: the last TIMED_LAUNCHES of the process
            ft, wt = f[-TIMED_LAUNCHES:], w[-TIMED_LAUNCHES:]
            out["kernels"][name].update({
                "launches": TIMED_LAUNCHES, "launches_in_process": n,
                "fetch_bytes_per_launch": sum(ft) * 1024 / TIMED_LAUNCHES,
                "write_bytes_per_launch": sum(wt) * 1024 / TIMED_LAUNCHES,
                "hbm_bytes_per_launch": (sum(ft) + sum(wt)) * 1024 / TIMED_LAUNCHES,
                "note": "averages over the last %d launches of the process = the timed region" % TIMED_LAUNCHES})
        if "ped_accumulate" in name:
            tot_b += sum(f) * 1024 + sum(w) * 1024
            tot_n += n
    out["accumulate_kernels"] = {"launches": tot_n, "hbm_bytes_per_launch": tot_b / max(tot_n, 1)}
    name = sys.argv[3] if len(sys.argv) > 3 else "r02_pmc_traffic.json"
    if len(sys.argv) > 4:
        out["config_key"] = sys.argv[4]
    if len(sys.argv) > 5:
        out["config"] = sys.argv[5]
    json.dump(out, open("profiles/" + name, "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Summarises two rocprofv3 PMC passes of bench.py (one with FETCH_SIZE, one with WRITE_SIZE) into
profiles/r01_pmc_traffic.json.  Run on the GPU box:

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
    acc = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        name = r["Kernel_Name"].split("(")[0].replace("void ", "")
        acc[name][0] += float(r["Counter_Value"])
        acc[name][1] += 1
    return acc


def main():
    fetch = per_kernel(sys.argv[1], "FETCH_SIZE")
    write = per_kernel(sys.argv[2], "WRITE_SIZE")
    out = {"units": "bytes; FETCH_SIZE KiB x 1024 (calibrated on the 64-B gathers, see docstring), WRITE_SIZE KiB x 1024",
           "kernels": {}}
    tot_b, tot_n = 0.0, 0
    for name in sorted(set(fetch) | set(write)):
        f, fn = fetch.get(name, [0.0, 0])
        w, wn = write.get(name, [0.0, 0])
        n = max(fn, wn, 1)
        b = (f * 1024 + w * 1024) / n
        out["kernels"][name] = {"launches": n, "fetch_bytes_per_launch": f * 1024 / n,
                                "write_bytes_per_launch": w * 1024 / n, "hbm_bytes_per_launch": b}
        if "ped_accumulate" in name:
            tot_b += f * 1024 + w * 1024
            tot_n += n
    out["accumulate_kernels"] = {"launches": tot_n, "hbm_bytes_per_launch": tot_b / max(tot_n, 1)}
    json.dump(out, open("profiles/r01_pmc_traffic.json", "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Identity of the code an evidence pass ran on (the staleness guard): sha256 of the loaded library
and ONE sha256 over the kernel sources (csrc/*.hip, *.hpp, Makefile, include/starkperp.h, in name order).
tests/test_evidence_fresh_cpu.py recomputes the source hash from the working tree and refuses an evidence set that was
collected before the last source change.

    python tools/evidence_stamp.py            # prints the JSON object
    from tools.evidence_stamp import source_hash"""
import glob
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "stark-perpetual_amd", "csrc")


def source_files():
    files = sorted(glob.glob(os.path.join(CSRC, "*.hip")) + glob.glob(os.path.join(CSRC, "*.hpp")))
    return files + [os.path.join(CSRC, "Makefile"), os.path.join(ROOT, "include", "starkperp.h")]


def source_hash():
    h = hashlib.sha256()
    for f in source_files():
        h.update(os.path.relpath(f, ROOT).encode() + b"\0")
        h.update(open(f, "rb").read())
        h.update(b"\0")
    return h.hexdigest()


def lib_hash(path=None):
    path = path or os.environ.get("STARKPERP_LIB") or os.path.join(ROOT, "stark-perpetual_amd", "lib", "libstarkperp.so")
    h = hashlib.sha256()
    with open(path, "rb") as f:
        for blk in iter(lambda: f.read(1 << 20), b""):
            h.update(blk)
    return h.hexdigest()


def main():
    out = {"round": 6, "collected_utc": time.strftime("%Y-%m-%dT%H:%M:%SZ", time.gmtime()),
           "csrc_sha256": source_hash(), "csrc_files": [os.path.relpath(f, ROOT) for f in source_files()],
           "lib_sha256": lib_hash(), "bench_py_sha16": hashlib.sha256(open(os.path.join(ROOT, "bench.py"), "rb").read()).hexdigest()[:16],
           "script": "tools/run_r06_prof.sh"}
    json.dump(out, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Height-64 sparse tree update of 4096 leaves on existing state (configs[2]'s tree half), host-inclusive
timing of the NumPy entry point; run under rocprofv3 --kernel-trace --stats for the kernel breakdown."""
import os, random, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "stark-perpetual_amd"), os.path.join(ROOT, "tests")]
import numpy as np
from starkperp import _lib, state, batch_np as bn
lib = _lib.ensure_init(0, int(sys.argv[1]) if len(sys.argv) > 1 else 26)
rng = random.Random(77)
tree = state.LibrarySparseTree(64, 0)
def batch():
    keys = np.array([rng.randrange(2**64) for _ in range(4096)], dtype=np.uint64)
    leaves = bn.felts_from_ints([rng.randrange(1, 2**64) for _ in range(4096)])
    return keys, leaves
tree.update_arrays(*batch())
ts = []
for _ in range(6):
    k, l = batch()
    t0 = time.perf_counter()
    tree.update_arrays(k, l)
    ts.append((time.perf_counter() - t0) * 1e3)
print("update of 4096 leaves at height 64 on existing state, ms:", " ".join("%.2f" % t for t in ts))
tree.close()

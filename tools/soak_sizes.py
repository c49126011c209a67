#!/usr/bin/env python3
"""Soak of every hash-kernel variant: batches of log-uniform random sizes 1 .. 2^17 (quad kernels with 8 / 4 / 2
quads, split kernels with 8 / 4 / 2 / 1 lanes, fused and unfused, bulk + batched inversion), edge values mixed
in, GPU vs the optimised C comparator (itself pinned by the reference goldens).
    python tools/soak_sizes.py [batches=300]"""
import math, os, sys, time, random
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "stark-perpetual_amd")):
    sys.path.insert(0, p)
from oracle import cref
from starkperp import batch_np as bn, _lib

P = 2**251 + 17 * 2**192 + 1
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
rng = random.Random(20260930)
edge = [0, 1, 2, P - 1, P - 2, 2**251, 2**250, 2**192, 2**29 - 1, (P - 1) // 2]
bad = total = 0
sizes = {}
t0 = time.time()
for it in range(reps):
    n = max(1, int(2 ** rng.uniform(0, 17)))
    if it < 40:
        n = [1, 2, 3, 7, 8, 9, 63, 64, 65, 255, 256, 257, 1023, 1024, 1025, 2047, 2048, 2049, 4095, 4096, 4097, 8191, 8192,
             8193, 16383, 16384, 16385, 32767, 32768, 32769, 65535, 65536, 65537, 81920, 131071, 131072, 100000, 5, 33, 513][it]
    xs = [rng.choice(edge) if rng.random() < 0.02 else rng.randrange(P) for _ in range(n)]
    ys = [rng.choice(edge) if rng.random() < 0.02 else rng.randrange(P) for _ in range(n)]
    exp, st = cref.opt_pedersen_hash_many(xs, ys)
    got = bn.ints_from_felts(bn.pedersen_hash_many(bn.felts_from_ints(xs), bn.felts_from_ints(ys)))
    bad += sum(1 for a, b in zip(exp, got) if a != b) + sum(st)
    total += n
print("window bits %d: %d batches, %d hashes compared, %d mismatches, %.1f s" % (
    _lib.ensure_init().sp_window_bits(), reps, total, bad, time.time() - t0))

#!/usr/bin/env python3
"""Small-batch latency of the host-pointer entry points (development aid)."""
import os, sys, time, random
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "stark-perpetual_amd"))
from starkperp import batch
N = batch.EC_ORDER
rng = random.Random(1)
for n in (1, 8, 64, 512):
    ds = [rng.randrange(1, N) for _ in range(n)]
    zs = [rng.randrange(2**251) for _ in range(n)]
    pubs = batch.public_keys_many(ds)
    sigs = batch.sign_many(zs, ds)
    rs, ss = [a for a, _ in sigs], [b for _, b in sigs]
    xs = [q[0] for q in pubs]
    for label, fn in (("verify ladder", lambda: batch.verify_codes(zs, rs, ss, xs, key_tables=False)),
                      ("verify tables", lambda: batch.verify_codes(zs, rs, ss, xs, key_tables=True)),
                      ("public keys", lambda: batch.public_keys_many(ds)),
                      ("pedersen", lambda: batch.pedersen_hash_many(zs, zs))):
        fn()
        t0 = time.perf_counter()
        for _ in range(5):
            out = fn()
        dt = (time.perf_counter() - t0) / 5
        print("n=%4d %-14s %.3f ms" % (n, label, dt * 1e3))

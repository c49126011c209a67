#!/usr/bin/env python3
"""Times the legs of the key-table soak separately: signing, the C oracle, registration + keyed verification
(point keys, x-only keys), the ladder.   python tools/quick_keyreg.py [n_sigs=131072]"""
import os, sys, time, random
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "stark-perpetual_amd")):
    sys.path.insert(0, p)
from oracle import cref
from starkperp import batch

N = batch.EC_ORDER
nsig = int(sys.argv[1]) if len(sys.argv) > 1 else 131072
rng = random.Random(5)
nk = max(1, nsig // 8)
ds = [rng.randrange(1, N) for _ in range(nk)]
t0 = time.time(); pubs = batch.public_keys_many(ds); print("public keys %.2f s" % (time.time() - t0), flush=True)
own = [rng.randrange(nk) for _ in range(nsig)]
zs = [rng.randrange(2**251) for _ in range(nsig)]
t0 = time.time(); sig = batch.sign_many(zs, [ds[o] for o in own]); print("sign_many %.2f s" % (time.time() - t0), flush=True)
rs, ss = [a for a, _ in sig], [b for _, b in sig]
keys = [pubs[o] for o in own]
for label, ks, kt in (("tables, point keys, first call", keys, True), ("tables, point keys, second call", keys, True),
                      ("tables, x-only keys, first call", [q[0] for q in keys], True),
                      ("tables, x-only keys, second call", [q[0] for q in keys], True),
                      ("ladder, x-only keys", [q[0] for q in keys], False), ("policy, x-only keys", [q[0] for q in keys], None)):
    t0 = time.time(); got = batch.verify_codes(zs, rs, ss, ks, key_tables=kt)
    print("%-34s %.2f s  true=%d" % (label, time.time() - t0, got.count(1)), flush=True)
if len(sys.argv) > 2:
    t0 = time.time(); exp = cref.verify_codes(zs, rs, ss, keys); print("C oracle %.2f s true=%d" % (time.time() - t0, exp.count(1)))

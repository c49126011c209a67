#!/usr/bin/env python3
"""One-off soak: N random Pedersen hashes and M random signatures, GPU vs the C oracle.
    python tools/soak.py [log2_hashes=20] [n_sigs=16384]"""
import os, sys, time, random
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "stark-perpetual_amd")):
    sys.path.insert(0, p)
from oracle import cref
from starkperp import batch

P, N = batch.FIELD_PRIME, batch.EC_ORDER
logn = int(sys.argv[1]) if len(sys.argv) > 1 else 20
nsig = int(sys.argv[2]) if len(sys.argv) > 2 else 16384
rng = random.Random(20260929)
n = 1 << logn
bad = 0
t0 = time.time()
for chunk in range(0, n, 1 << 17):
    m = min(1 << 17, n - chunk)
    xs = [rng.randrange(P) for _ in range(m)]
    ys = [rng.randrange(P) for _ in range(m)]
    exp, st = cref.pedersen_hash_many(xs, ys)
    got = batch.pedersen_hash_many(xs, ys)
    bad += sum(1 for a, b in zip(exp, got) if a != b) + sum(st)
print("hashes: %d compared, %d mismatches, %.1f s" % (n, bad, time.time() - t0))
t0 = time.time()
ds = [rng.randrange(1, N) for _ in range(nsig)]
zs = [rng.randrange(2**251) for _ in range(nsig)]
ks = [rng.randrange(1, N) for _ in range(nsig)]
pubs = batch.public_keys_many(ds)
rs, ss, st = batch.sign_attempt_many(zs, ds, ks)
for i in range(0, nsig, 2):
    ss[i] = (ss[i] * 3 + 1) % (N - 1) + 1
exp = cref.verify_codes(zs, rs, ss, pubs)
got_p = batch.verify_codes(zs, rs, ss, pubs)
got_x = batch.verify_codes(zs, rs, ss, [q[0] for q in pubs])
print("signatures: %d, oracle true=%d, point-key mismatches=%d, x-only mismatches=%d, %.1f s" % (
    nsig, exp.count(1), sum(a != b for a, b in zip(exp, got_p)), sum(a != b for a, b in zip(exp, got_x)),
    time.time() - t0))
# key tables (signed comb) against the same oracle verdicts, keys repeated 8x
t0 = time.time()
nk = max(1, nsig // 8)
own = [rng.randrange(nk) for _ in range(nsig)]
zs2 = [rng.randrange(2**251) for _ in range(nsig)]
sig2 = batch.sign_many(zs2, [ds[o] for o in own])
rs2, ss2 = [a for a, _ in sig2], [b for _, b in sig2]
keys2 = [pubs[o] for o in own]
for i in range(0, nsig, 3):
    zs2[i] = (zs2[i] + 1 + rng.randrange(1000)) % 2**251
exp2 = cref.verify_codes(zs2, rs2, ss2, keys2)
tab_p = batch.verify_codes(zs2, rs2, ss2, keys2, key_tables=True)
tab_x = batch.verify_codes(zs2, rs2, ss2, [q[0] for q in keys2], key_tables=True)
lad_x = batch.verify_codes(zs2, rs2, ss2, [q[0] for q in keys2], key_tables=False)
print("key tables: %d signatures over %d keys, oracle true=%d, point-key mismatches=%d, x-only vs oracle=%d, "
      "x-only tables vs ladder=%d, %.1f s" % (
          nsig, nk, exp2.count(1), sum(a != b for a, b in zip(exp2, tab_p)),
          sum(a != b for a, b in zip(exp2, tab_x)), sum(a != b for a, b in zip(tab_x, lad_x)), time.time() - t0))

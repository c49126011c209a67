#!/usr/bin/env python3
"""A single 4000-word hash chain folded from the right (the program-hash shape) and from the left, host-inclusive:
ped_chain_kernel against one launch per hash (STARKPERP_NO_CHAIN_FUSION=1).  python tools/quick_chain.py"""
import os, sys, time, random
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "stark-perpetual_amd"))
from starkperp import hash_chains as hc, batch
rng = random.Random(1); P = 2**251 + 17 * 2**192 + 1
w = [rng.randrange(P) for _ in range(4000)]
hc.compute_hash_chain(w[:10]); batch.pedersen_chain(w[:10])
for name, fn in (("right fold (compute_hash_chain)", hc.compute_hash_chain), ("left fold (pedersen_chain)", batch.pedersen_chain)):
    t0 = time.perf_counter(); r = fn(w); dt = time.perf_counter() - t0
    print("%s of 4000 words: %.1f ms = %.2f us per hash (fusion %s) -> %s" % (name, dt * 1e3, dt * 1e6 / 3999, "off" if os.environ.get("STARKPERP_NO_CHAIN_FUSION") else "on", hex(r)[:12]))

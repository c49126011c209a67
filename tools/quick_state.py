#!/usr/bin/env python3
"""Persistent sparse-tree update timing (development aid)."""
import os, sys, time, random
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "stark-perpetual_amd"))
from starkperp import state, batch
rng = random.Random(5)
n, height = int(sys.argv[1]) if len(sys.argv) > 1 else 4096, 64
P = batch.FIELD_PRIME
t = state.SparseMerkleTree(height)
for it in range(3):
    mods = {rng.randrange(2**height): rng.randrange(P) for _ in range(n)}
    t0 = time.perf_counter(); old, new = t.update(mods); dt = time.perf_counter() - t0
    print("python tree, batch %d: %.1f ms (root %x...)" % (it, dt * 1e3, new >> 200))
if hasattr(state, "LibrarySparseTree"):
    rng = random.Random(5)
    d = state.LibrarySparseTree(height)
    for it in range(3):
        mods = {rng.randrange(2**height): rng.randrange(P) for _ in range(n)}
        t0 = time.perf_counter(); old, new = d.update(mods); dt = time.perf_counter() - t0
        print("library tree, batch %d: %.1f ms (root %x...)" % (it, dt * 1e3, new >> 200))

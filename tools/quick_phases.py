#!/usr/bin/env python3
"""Prover phase timings at 2^20 rows (dev aid): LDE, AIR evaluation, first FRI fold, commits, whole job."""
import os, sys, time, random
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "stark-perpetual_amd"))
import torch
from starkperp import _lib, stark
wb = int(sys.argv[1]) if len(sys.argv) > 1 else 26
log_rows = int(sys.argv[2]) if len(sys.argv) > 2 else 20
lib = _lib.ensure_init(0, wb)
m = 1 << (log_rows - 9)
g = torch.Generator().manual_seed(1)
def felts(n):
    t = torch.randint(-(2**63), 2**63 - 1, (n, 4), dtype=torch.int64, generator=g); t[:, 3] &= (1 << 58) - 1
    return t.cuda()
xs, ys = felts(m), felts(m)
rng = random.Random(2)
P = stark.FIELD_PRIME
alphas = [rng.randrange(P) for _ in range(stark.N_CONSTRAINTS)]
betas = [rng.randrange(P) for _ in range(log_rows + 2 - 6)]
trace = stark.pedersen_trace(xs, ys)
n = 512 * m
per = stark.periodic_lde(n)
def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps
t_lde = stark.lde(trace)
comp = stark.air_eval(t_lde, per, n, alphas)
n_lde = 4 * n
res = {
  "lde": (timed(lambda: stark.lde(trace)), 4 * (2 * 64 * n + 32 * n + 32 * n_lde + 2 * 64 * n_lde)),
  "air": (timed(lambda: stark.air_eval(t_lde, per, n, alphas)), (7 + 6 + 1) * 32 * n_lde),
  "fold": (timed(lambda: stark.fri_fold(comp, betas[0], stark.FIELD_GEN)), (32 + 16) * n_lde),
  "commit_trace": (timed(lambda: stark.commit_rows(t_lde), 2), 0),
  "commit_comp": (timed(lambda: stark.commit_rows(comp.unsqueeze(0)), 2), 0),
}
for k, (t, b) in res.items():
    print("%-14s %8.3f ms  %s" % (k, t * 1e3, ("%.2f TB/s = %.3f of 8" % (b / t / 1e12, b / t / 8e12)) if b else ""))
def job():
    return stark.prove_commitments(xs, ys, alphas, betas)
print("job %.2f ms" % (timed(job, 3) * 1e3))

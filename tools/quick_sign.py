#!/usr/bin/env python3
"""Device-resident signer by batch size: RFC 6979 nonce + attempt (sp_ecdsa_sign_rfc6979_batch_dev), one attempt with
caller nonces (sp_ecdsa_sign_batch_dev) and d * G alone (sp_public_key_batch_dev) - which part of a signature
costs what (dev aid).    python tools/quick_sign.py"""
import os, sys, random
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "stark-perpetual_amd"))
import torch
from starkperp import _lib, batch, stark as st
lib = _lib.ensure_init(0, int(os.environ.get("STARKPERP_WINDOW_BITS", "26")))


def timed(fn, iters=5):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters / 1e3


def felts(n, seed, bound):
    g = torch.Generator().manual_seed(seed)
    t = torch.randint(-(2**63), 2**63 - 1, (n, 4), dtype=torch.int64, generator=g)
    t[:, 3] &= bound
    t[:, 0] |= 1
    return t.cuda()


for log_n in (12, 14, 16, 18, 20):
    n = 1 << log_n
    z, d, k = felts(n, 1, (1 << 58) - 1), felts(n, 2, (1 << 58) - 1), felts(n, 3, (1 << 58) - 1)
    t_full = timed(lambda: batch.sign_dev(z, d))
    t_att = timed(lambda: batch.sign_dev(z, d, k=k))
    t_pub = timed(lambda: batch.public_keys_dev(d, want_y=False))
    print("n = 2^%-2d  rfc6979 + attempt %8.3f ms %.3e/s   attempt (caller nonce) %8.3f ms %.3e/s   d*G %8.3f ms %.3e/s"
          % (log_n, t_full * 1e3, n / t_full, t_att * 1e3, n / t_att, t_pub * 1e3, n / t_pub), flush=True)

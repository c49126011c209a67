#!/usr/bin/env python3
"""Device-resident ECDSA verification rates at one batch size (default 2^16, the size bench.py quotes):
per-signature ladder (x-only keys) and per-key comb tables.  `python tools/quick_ecdsa.py [log_n] [iters]`"""
import os
import random
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "stark-perpetual_amd"))
import numpy as np
import torch
from starkperp import _lib, batch, stark


def main():
    log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    iters = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    lib = _lib.ensure_init()
    dev = "cuda"
    stream = torch.cuda.current_stream().cuda_stream
    nv = 1 << log_n
    rng = random.Random(21)
    dsk = [rng.randrange(1, batch.EC_ORDER) for _ in range(nv)]
    zv = [rng.randrange(2**251) for _ in range(nv)]
    kv = [rng.randrange(1, batch.EC_ORDER) for _ in range(nv)]
    pv = batch.public_keys_many(dsk)
    rv, sv, stv = batch.sign_attempt_many(zv, dsk, kv)
    dz, dr, dsig, dq = (stark.felts_to_tensor(v, dev) for v in (zv, rv, sv, [q[0] for q in pv]))
    res = torch.zeros(nv, dtype=torch.uint8, device=dev)

    def timed(fn):
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / iters / 1e3

    t = timed(lambda: _lib.check(lib.sp_ecdsa_verify_batch_dev(
        dz.data_ptr(), dr.data_ptr(), dsig.data_ptr(), dq.data_ptr(), None, res.data_ptr(), nv, stream), "verify"))
    ok = int((res == 1).sum()) == stv.count(0)
    print("ladder  n=2^%d  %.1f us  %.3e /s  all_true=%s" % (log_n, t * 1e6, nv / t, ok))
    batch.key_cache_reset()
    t0 = time.perf_counter()
    slots = batch.register_keys([q[0] for q in pv])
    t_reg = time.perf_counter() - t0
    dslots = torch.from_numpy(np.asarray(slots, dtype=np.uint32).view(np.int32)).to(dev)
    t = timed(lambda: _lib.check(lib.sp_ecdsa_verify_keyed_dev(
        dz.data_ptr(), dr.data_ptr(), dsig.data_ptr(), dslots.data_ptr(), res.data_ptr(), nv, stream), "keyed"))
    ok = int((res == 1).sum()) == stv.count(0)
    print("keyed   n=2^%d  %.1f us  %.3e /s  all_true=%s   registration %.3e keys/s (host-inclusive)" % (
        log_n, t * 1e6, nv / t, ok, nv / t_reg))


if __name__ == "__main__":
    main()

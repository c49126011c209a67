#!/usr/bin/env python3
"""Ad-hoc timing of the device entry points (development aid, not the contract bench)."""
import ctypes
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "stark-perpetual_amd"))
import torch  # noqa: E402

from starkperp import _lib  # noqa: E402

P = 2**251 + 17 * 2**192 + 1


def rand_felts(n, seed):
    g = torch.Generator(device="cpu").manual_seed(seed)
    t = torch.randint(-(2**63), 2**63 - 1, (n, 4), dtype=torch.int64, generator=g)
    t[:, 3] &= (1 << 58) - 1  # < 2^250 < p
    return t.cuda()


def timeit(fn, iters=5):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    wb = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    t0 = time.time()
    lib = _lib.ensure_init(0, wb)
    print("init %.2fs window_bits=%d table=%.1f MiB" % (time.time() - t0, lib.sp_window_bits(),
                                                       lib.sp_table_bytes() / 2**20))
    st = torch.cuda.current_stream().cuda_stream
    for logn in (10, 14, 16, 18, 20, 22):
        n = 1 << logn
        x, y = rand_felts(n, 1), rand_felts(n, 2)
        out = torch.empty_like(x)
        ms = timeit(lambda: _lib.check(lib.sp_pedersen_batch_dev(x.data_ptr(), y.data_ptr(), out.data_ptr(),
                                                                 None, n, st), "ped"))
        print("pedersen batch n=2^%d: %.3f ms  %.3e hashes/s" % (logn, ms, n / ms * 1e3))
    for h in (10, 16, 20):
        n = 1 << h
        lv = torch.zeros((2 * n - 1, 4), dtype=torch.int64, device="cuda")
        lv[:n] = rand_felts(n, 3)
        ms = timeit(lambda: _lib.check(lib.sp_merkle_build_dev(lv.data_ptr(), h, None, st), "merkle"))
        print("merkle rebuild h=%d: %.3f ms  %.3e hashes/s" % (h, ms, (n - 1) / ms * 1e3))


if __name__ == "__main__":
    main()

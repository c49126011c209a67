#!/usr/bin/env python3
"""Does a NEW HIP stream stall once after a fixed number of operations?  (Round 6: one sp_order_batch call among the first
ten of a process takes 7 - 15 ms inside the runtime's enqueue on the tree's own stream, at a fixed operation count.)
Enqueue `n` tiny operations on a fresh stream, host-time each enqueue, print the slow ones.
    python tools/ubench/stream_warmup.py [n=1200] [kind=kernel|memset|copy|copy128k|mixed]"""
import sys
import time

import torch


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1200
    kind = sys.argv[2] if len(sys.argv) > 2 else "kernel"
    dev = torch.device("cuda", 0)
    x = torch.zeros(1024, device=dev)
    h = torch.zeros(1024).pin_memory()
    xb = torch.zeros(32768, device=dev)          # 128 KiB
    hb = torch.zeros(32768).pin_memory()
    hp = torch.zeros(32768)                      # pageable
    torch.cuda.synchronize()
    for trial in range(3):
        st = torch.cuda.Stream(device=dev)
        slow, total = [], 0.0
        with torch.cuda.stream(st):
            for i in range(n):
                t0 = time.perf_counter()
                if kind == "kernel":
                    x.add_(1.0)
                elif kind == "memset":
                    x.zero_()
                elif kind == "copy":
                    x.copy_(h, non_blocking=True)
                elif kind == "copy128k":
                    xb.copy_(hb, non_blocking=True)
                else:  # mixed: what one sp_order_batch call does to the copy engines
                    xb.copy_(hb, non_blocking=True)
                    hb.copy_(xb, non_blocking=True)
                    xb.copy_(hp, non_blocking=True)
                    xb.add_(1.0)
                if i % 32 == 31:
                    st.synchronize()  # like an update: a burst of operations, then a wait
                dt = time.perf_counter() - t0
                total += dt
                if dt > 1e-3:
                    slow.append((i, round(dt * 1e3, 2)))
        st.synchronize()
        print("trial %d: %d %s operations on a fresh stream, %.1f ms in all; enqueues above 1 ms (index, ms): %s" % (
            trial, n, kind, total * 1e3, slow))


if __name__ == "__main__":
    main()

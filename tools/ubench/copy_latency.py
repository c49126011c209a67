#!/usr/bin/env python3
"""Host <-> device copy latency for the sizes on sp_order_batch's critical path (round 6): pageable against pinned host
memory, hipMemcpyAsync + hipStreamSynchronize per copy, through torch (which calls the same runtime entry points).
    python tools/ubench/copy_latency.py"""
import time

import torch


def bench(src, dst, iters=300):
    st = torch.cuda.current_stream()
    for _ in range(20):
        dst.copy_(src, non_blocking=True)
        st.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        dst.copy_(src, non_blocking=True)
        st.synchronize()
    return (time.perf_counter() - t0) / iters * 1e6


def main():
    dev = torch.device("cuda", 0)
    print("%10s  %12s %12s %12s %12s" % ("bytes", "H2D pageable", "H2D pinned", "D2H pageable", "D2H pinned"))
    for nbytes in (2048, 32768, 131072, 524288, 2 << 20):
        host = torch.empty(nbytes, dtype=torch.uint8)
        pinned = torch.empty(nbytes, dtype=torch.uint8).pin_memory()
        d = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        print("%10d  %9.1f us %9.1f us %9.1f us %9.1f us" % (
            nbytes, bench(host, d), bench(pinned, d), bench(d, host), bench(d, pinned)))
    # a host memcpy of the same sizes (what staging through a pinned buffer adds)
    a, b = torch.empty(131072, dtype=torch.uint8), torch.empty(131072, dtype=torch.uint8)
    t0 = time.perf_counter()
    for _ in range(2000):
        b.copy_(a)
    print("host memcpy of 131072 bytes: %.2f us" % ((time.perf_counter() - t0) / 2000 * 1e6))


if __name__ == "__main__":
    main()

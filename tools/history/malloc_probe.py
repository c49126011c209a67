"""Does freeing and re-allocating 75 GiB within milliseconds stall by itself?  hipMalloc / hipMemset(first MiB) / hipFree
cycles through ctypes, no kernels of ours (profiles/r04_table_build.txt, the re-initialisation outlier)."""
import ctypes, time, torch
torch.zeros(1, device="cuda")
hip = ctypes.CDLL("libamdhip64.so")
n = 75 << 30
for rep in range(8):
    p = ctypes.c_void_p()
    t = time.perf_counter(); rc = hip.hipMalloc(ctypes.byref(p), ctypes.c_size_t(n)); a = time.perf_counter() - t
    t = time.perf_counter(); hip.hipMemset(p, 0, ctypes.c_size_t(n)); hip.hipDeviceSynchronize(); b = time.perf_counter() - t
    t = time.perf_counter(); hip.hipFree(p); c = time.perf_counter() - t
    print("rep %d rc=%d: hipMalloc %.3f s, memset of all 75 GiB %.3f s, hipFree %.3f s" % (rep, rc, a, b, c), flush=True)

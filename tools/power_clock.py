#!/usr/bin/env python3
"""Power and clock telemetry under the bulk hash kernel: 2^22 independent hashes in a loop for ~10 s while
`rocm-smi` is sampled from a side thread (board power, sclk, temperature).  Evidence for the clock the VALU-issue
fractions are priced against (peak at the nominal 2.4 GHz; the chip holds less under this kernel).
    python tools/power_clock.py [window_bits=26] [seconds=10] [few<k>]"""
import json, os, subprocess, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "stark-perpetual_amd"))
import torch
from starkperp import _lib

wb = int(sys.argv[1]) if len(sys.argv) > 1 else 26
secs = float(sys.argv[2]) if len(sys.argv) > 2 else 10.0
lib = _lib.ensure_init(0, wb)
n = 1 << 22
g = torch.Generator().manual_seed(1)
def felts():
    t = torch.randint(-(2**63), 2**63 - 1, (n, 4), dtype=torch.int64, generator=g); t[:, 3] &= (1 << 58) - 1
    return t.cuda()
x, y = felts(), felts(); o = torch.empty_like(x)
if len(sys.argv) > 3 and sys.argv[3].startswith("few"):  # few<k>: 2^k distinct pairs repeated - gathers served by L2 / MALL
    idx = torch.arange(n, device="cuda") % (1 << int(sys.argv[3][3:] or 10))
    x, y = x[idx].contiguous(), y[idx].contiguous()
s = torch.cuda.current_stream().cuda_stream
samples, stop = [], threading.Event()

def smi(tag):
    try:
        out = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--showtemp", "--json"], capture_output=True,
                             text=True, timeout=20).stdout
        samples.append((tag, time.time(), json.loads(out)))
    except Exception as e:  # noqa: BLE001
        samples.append((tag, time.time(), {"error": repr(e)}))

def sampler():
    while not stop.is_set():
        smi("load")
        stop.wait(0.5)

smi("idle")
th = threading.Thread(target=sampler); th.start()
t0 = time.time(); calls = 0
while time.time() - t0 < secs:
    for _ in range(16):
        _lib.check(lib.sp_pedersen_batch_dev(x.data_ptr(), y.data_ptr(), o.data_ptr(), None, n, s), "ped")
    torch.cuda.synchronize(); calls += 16
dt = time.time() - t0
stop.set(); th.join()
time.sleep(2.0); smi("idle_after")
print("bulk: %d calls of 2^22 hashes in %.2f s = %.3e hashes/s (window bits %d)" % (calls, dt, calls * n / dt, wb))
for tag, t, d in samples:
    card = d.get("card0", d)
    keep = {k: v for k, v in card.items() if any(w in k.lower() for w in ("power", "sclk", "mclk", "fclk", "temperature (sensor junction", "temperature (sensor edge", "error"))}
    print("%-10s t=%6.2f s  %s" % (tag, t - t0, json.dumps(keep)))

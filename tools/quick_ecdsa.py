#!/usr/bin/env python3
"""Ad-hoc ECDSA throughput (development aid)."""
import os, sys, time, random, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "stark-perpetual_amd"))
import torch
from starkperp import _lib, batch, stark

N = batch.EC_ORDER
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 16
rng = random.Random(1)
ds = [rng.randrange(1, N) for _ in range(n)]
zs = [rng.randrange(2**251) for _ in range(n)]
ks = [rng.randrange(1, N) for _ in range(n)]
t0 = time.time(); pubs = batch.public_keys_many(ds); t1 = time.time()
print("public keys: %.3f s host-inclusive (%d)" % (t1 - t0, n))
t0 = time.time(); rs, ss, st = batch.sign_attempt_many(zs, ds, ks); t1 = time.time()
print("sign attempts: %.3f s host-inclusive, ok=%d" % (t1 - t0, st.count(0)))
lib = _lib.ensure_init()
dz, dr, dss = (stark.felts_to_tensor(v) for v in (zs, rs, ss))
qx = stark.felts_to_tensor([p[0] for p in pubs]); qy = stark.felts_to_tensor([p[1] for p in pubs])
res = torch.zeros(n, dtype=torch.uint8, device="cuda")
s = torch.cuda.current_stream().cuda_stream
for name, py in (("x-only", None), ("point", qy.data_ptr())):
    for _ in range(2):
        torch.cuda.synchronize(); t0 = time.time()
        _lib.check(lib.sp_ecdsa_verify_batch_dev(dz.data_ptr(), dr.data_ptr(), dss.data_ptr(), qx.data_ptr(), py, res.data_ptr(), n, s), "verify")
        torch.cuda.synchronize(); t1 = time.time()
    ok = int((res == 1).sum())
    print("verify %s: %.3f ms -> %.3e verifies/s (true=%d of %d signed ok=%d)" % (name, (t1 - t0) * 1e3, n / (t1 - t0), ok, n, st.count(0)))

# key tables: registration cost and warm verification rate
import numpy as np
for label, keyset in (("x-only", [p[0] for p in pubs]), ("point", pubs)):
    batch.key_cache_reset()
    t0 = time.time(); slots = batch.register_keys(keyset); t1 = time.time()
    print("register %d %s keys: %.1f ms host-inclusive (%.3e keys/s)" % (n, label, (t1 - t0) * 1e3, n / (t1 - t0)))
    dslots = torch.from_numpy(np.asarray(slots, dtype=np.uint32).view(np.int32)).cuda()
    for _ in range(3):
        torch.cuda.synchronize(); t0 = time.time()
        _lib.check(lib.sp_ecdsa_verify_keyed_dev(dz.data_ptr(), dr.data_ptr(), dss.data_ptr(), dslots.data_ptr(), res.data_ptr(), n, s), "keyed")
        torch.cuda.synchronize(); t1 = time.time()
    print("verify keyed %s: %.3f ms -> %.3e verifies/s (true=%d)" % (label, (t1 - t0) * 1e3, n / (t1 - t0), int((res == 1).sum())))

# full signing (RFC 6979 nonce + attempt on the device) vs the host-nonce path on a sample
t0 = time.time(); sigs = batch.sign_many(zs, ds); t1 = time.time()
print("sign_many (device RFC 6979): %d in %.1f ms host-inclusive -> %.3e signatures/s" % (n, (t1 - t0) * 1e3, n / (t1 - t0)))
m = min(n, 2048)
t0 = time.time(); ref = batch._sign_many_host_nonces(zs[:m], ds[:m], [None] * m); t1 = time.time()
print("host nonces: %d in %.1f ms -> %.3e signatures/s; equal=%s" % (m, (t1 - t0) * 1e3, m / (t1 - t0), ref == sigs[:m]))

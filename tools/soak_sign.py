#!/usr/bin/env python3
"""Self-consistency soak of the device signer at 2^20 items (compacted RFC 6979 pipeline): every signature verifies on
the x-only ladder against the public key derived on the device, none of the corrupted copies does; a sha256 of all (r, s)
is printed so that runs under different signer settings (STARKPERP_SIGN_COMPACT_MIN=0, STARKPERP_SIGN_CHUNK=..,
STARKPERP_SIGN_MASKED=1) can be compared.    python tools/soak_sign.py [items=1048576]"""
import hashlib, os, sys, torch
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, os.path.join(ROOT, "stark-perpetual_amd"))
from starkperp import _lib, batch
lib = _lib.ensure_init(0, 21)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
g = torch.Generator().manual_seed(2024)
def felts(bound):
    t = torch.randint(-(2**63), 2**63 - 1, (n, 4), dtype=torch.int64, generator=g); t[:, 3] &= bound; t[:, 0] |= 1
    return t.cuda()
z, d = felts((1 << 58) - 1), felts((1 << 58) - 1)
seeds = torch.randint(0, 2**62, (n,), dtype=torch.int64, generator=g); seeds[::2] = 0
r, s, st = batch.sign_dev(z, d, seeds=seeds.cuda())
qx, qy, pst = batch.public_keys_dev(d, want_y=True)
torch.cuda.synchronize()
assert int((st != 0).sum()) == 0 and int((pst != 0).sum()) == 0
res = torch.zeros(n, dtype=torch.uint8, device="cuda")
_lib.check(lib.sp_ecdsa_verify_batch_dev(z.data_ptr(), r.data_ptr(), s.data_ptr(), qx.data_ptr(), None, res.data_ptr(), n,
                                         torch.cuda.current_stream().cuda_stream), "verify")
torch.cuda.synchronize()
ok = int((res == 1).sum())
# corrupt: a different message must fail
z2 = z.clone(); z2[:, 0] ^= 2
_lib.check(lib.sp_ecdsa_verify_batch_dev(z2.data_ptr(), r.data_ptr(), s.data_ptr(), qx.data_ptr(), None, res.data_ptr(), n,
                                         torch.cuda.current_stream().cuda_stream), "verify")
torch.cuda.synchronize()
bad = int((res == 1).sum())
digest = hashlib.sha256(r.cpu().numpy().tobytes() + s.cpu().numpy().tobytes()).hexdigest()
print("%d signatures (half seeded): %d verify on the x-only ladder, %d of the corrupted copies do; sha256(r, s) = %s" % (n, ok, bad, digest))
assert ok == n and bad == 0

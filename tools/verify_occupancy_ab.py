#!/usr/bin/env python3
"""One library variant's verification rates for the occupancy A/B: device-resident ladder and
keyed verification at 2^16 .. 2^20 signatures, the 4096-verification ECDSA AIR evaluation (with a digest of its output, equal across variants).  The library is whatever STARKPERP_LIB names (default: the product library).

    STARKPERP_LIB=.../libstarkperp_v2.so python tools/verify_occupancy_ab.py [label]

Prints one line per measurement; tools/run_r06_verify_occupancy.sh runs it per variant into
profiles/r06_verify_occupancy.txt."""
import os
import random
import sys
import time

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "stark-perpetual_amd"))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
from starkperp import _lib, batch, stark  # noqa: E402


def timed(fn, iters):
    fn()
    torch.cuda.synchronize()
    best = None
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        t = e0.elapsed_time(e1) / iters / 1e3
        best = t if best is None else min(best, t)
    return best


def main():
    label = sys.argv[1] if len(sys.argv) > 1 else os.path.basename(_lib.LIB_PATH)
    lib = _lib.ensure_init()
    dev = "cuda"
    stream = torch.cuda.current_stream().cuda_stream
    print("== %s  (%s)" % (label, _lib.LIB_PATH))
    base = 1 << 16
    rng = random.Random(21)
    dsk = [rng.randrange(1, batch.EC_ORDER) for _ in range(base)]
    zv = [rng.randrange(2**251) for _ in range(base)]
    kv = [rng.randrange(1, batch.EC_ORDER) for _ in range(base)]
    pv = batch.public_keys_many(dsk)
    rv, sv, stv = batch.sign_attempt_many(zv, dsk, kv)
    good = stv.count(0)
    t1 = [stark.felts_to_tensor(v, dev) for v in (zv, rv, sv, [q[0] for q in pv])]
    batch.key_cache_reset()
    slots1 = torch.from_numpy(np.asarray(batch.register_keys([q[0] for q in pv]), dtype=np.uint32).view(np.int32)).to(dev)
    for log_n in (16, 17, 18, 20):
        rep = 1 << (log_n - 16)
        dz, dr, dsig, dq = (t.repeat(rep, 1).contiguous() for t in t1)
        dslots = slots1.repeat(rep).contiguous()
        nv = base * rep
        res = torch.zeros(nv, dtype=torch.uint8, device=dev)
        iters = max(1, 4 >> (log_n - 16))
        t = timed(lambda: _lib.check(lib.sp_ecdsa_verify_batch_dev(
            dz.data_ptr(), dr.data_ptr(), dsig.data_ptr(), dq.data_ptr(), None, res.data_ptr(), nv, stream), "verify"), iters)
        ok = int((res == 1).sum()) == good * rep
        print("ladder  2^%-2d  %9.1f us  %.4e sig/s  all_true=%s" % (log_n, t * 1e6, nv / t, ok))
        res.zero_()
        t = timed(lambda: _lib.check(lib.sp_ecdsa_verify_keyed_dev(
            dz.data_ptr(), dr.data_ptr(), dsig.data_ptr(), dslots.data_ptr(), res.data_ptr(), nv, stream), "keyed"), 4 * iters)
        ok = int((res == 1).sum()) == good * rep
        print("keyed   2^%-2d  %9.1f us  %.4e sig/s  all_true=%s" % (log_n, t * 1e6, nv / t, ok))
        del dz, dr, dsig, dq, dslots, res
    # the ECDSA AIR on 4096 verifications (2^22 trace rows, 2^24 LDE points): composition evaluation alone
    k = 4096
    idx = [i for i in range(base) if stv[i] == 0][:k]
    ws = [pow(sv[i], -1, batch.EC_ORDER) for i in idx]
    ok_w = [i for i, w in zip(idx, ws) if w < 2**251 and rv[i] < 2**251]
    if len(ok_w) < k:
        more = [i for i in range(base) if stv[i] == 0 and i not in set(idx)]
        for i in more:
            w = pow(sv[i], -1, batch.EC_ORDER)
            if w < 2**251 and rv[i] < 2**251:
                ok_w.append(i)
            if len(ok_w) == k:
                break
    sel = ok_w[:k]
    trace = stark.ecdsa_trace(*(stark.felts_to_tensor(v, dev) for v in (
        [zv[i] for i in sel], [rv[i] for i in sel], [pow(sv[i], -1, batch.EC_ORDER) for i in sel],
        [pv[i][0] for i in sel], [pv[i][1] for i in sel])))
    n = trace.shape[1]
    t_lde = stark.lde(trace)
    per = stark.periodic_lde(n, stark.FIELD_GEN, trace.device, "ecdsa")
    arng = random.Random(5)
    alphas = [arng.randrange(stark.FIELD_PRIME) for _ in range(stark.N_ECDSA_CONSTRAINTS)]
    t = timed(lambda: stark.air_eval(t_lde, per, n, alphas, stark.FIELD_GEN, "ecdsa"), 3)
    comp = stark.air_eval(t_lde, per, n, alphas, stark.FIELD_GEN, "ecdsa")
    digest = int(comp.to(torch.int64).sum().item()) & 0xFFFFFFFFFFFF
    print("air_eval_ecdsa  %d verifications (2^%d points)  %9.1f us  %.4e points/s  digest=%012x" % (
        k, (4 * n).bit_length() - 1, t * 1e6, 4 * n / t, digest))
    del trace, t_lde, per, comp
    # the signers (SP_SIGN_WAVES): 2^18 device-resident RFC 6979 signatures, 2^18 public keys
    rep = 4
    nv = base * rep
    dzs = t1[0].repeat(rep, 1).contiguous()
    dd = stark.felts_to_tensor(dsk, dev).repeat(rep, 1).contiguous()
    sr, ss = torch.zeros_like(dzs), torch.zeros_like(dzs)
    sst = torch.zeros(nv, dtype=torch.uint8, device=dev)
    t = timed(lambda: _lib.check(lib.sp_ecdsa_sign_rfc6979_batch_dev(
        dzs.data_ptr(), dd.data_ptr(), None, sr.data_ptr(), ss.data_ptr(), sst.data_ptr(), nv, stream), "sign"), 2)
    dg = (int(sr.sum().item()) ^ int(ss.sum().item())) & 0xFFFFFFFFFFFF
    print("sign_rfc6979  2^18  %9.1f us  %.4e sig/s  ok=%s digest=%012x" % (t * 1e6, nv / t, int((sst == 0).sum()) == nv, dg))
    qx, qy = torch.zeros_like(dzs), torch.zeros_like(dzs)
    t = timed(lambda: _lib.check(lib.sp_public_key_batch_dev(dd.data_ptr(), qx.data_ptr(), qy.data_ptr(), None, nv, stream), "pub"), 2)
    print("public_key    2^18  %9.1f us  %.4e keys/s  digest=%012x" % (t * 1e6, nv / t, int(qx.sum().item()) & 0xFFFFFFFFFFFF))
    del dzs, dd, sr, ss, sst, qx, qy
    # the Pedersen-step composition (SP_AIR_WAVES): 2^22 points of a 2^20-row trace
    from benchlib.common import seeded_felts
    m = 2048
    xs, ys = seeded_felts(torch, m, 11, dev), seeded_felts(torch, m, 12, dev)
    tr = stark.pedersen_trace(xs, ys)
    tl = stark.lde(tr)
    pr = stark.periodic_lde(512 * m, stark.FIELD_GEN, dev)
    al = [arng.randrange(stark.FIELD_PRIME) for _ in range(stark.N_CONSTRAINTS)]
    t = timed(lambda: stark.air_eval(tl, pr, 512 * m, al), 5)
    cm = stark.air_eval(tl, pr, 512 * m, al)
    print("air_eval (Pedersen AIR) 2^22 points  %9.1f us  digest=%012x" % (t * 1e6, int(cm.sum().item()) & 0xFFFFFFFFFFFF))
    del tr, tl, pr, cm
    # the bulk hash kernel (SP_ACC_WAVES): 2^22 independent hashes
    n = 1 << 22
    x, y = seeded_felts(torch, n, 7, dev), seeded_felts(torch, n, 8, dev)
    o = torch.empty_like(x)
    bulk = lambda: _lib.check(lib.sp_pedersen_batch_dev(x.data_ptr(), y.data_ptr(), o.data_ptr(), None, n, stream), "ped")  # noqa: E731
    timed(bulk, 20)  # pre-heat
    t = timed(bulk, 20)
    print("pedersen bulk 2^22 (w = %d)  %9.1f us  %.4e hashes/s  digest=%012x" % (
        int(lib.sp_window_bits()), t * 1e6, n / t, int(o.sum().item()) & 0xFFFFFFFFFFFF))


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""BASELINE.json configs[2] (4096 limit orders: message-hash chains -> keyed ECDSA verification -> height-64 orders-tree
update) as a measured pipeline.

    python tools/c3_probe.py calls [N]    N sp_order_batch calls on a tree that already holds state, every call's time
    python tools/c3_probe.py one          4 warm-up calls, a pause, then ONE call: under `rocprofv3 --kernel-trace`
                                          the kernels after the last long gap are that call (tools/c3_timeline.py)
    python tools/c3_probe.py pmc          the four kernels C3 is made of at known item counts, for a `--pmc
                                          SQ_INSTS_VALU SQ_WAVES` pass (tools/c3_counters.py): ladder and keyed
                                          verification of 2^16 signatures, one order batch (ped_chain / ped_path)
"""
import os
import random
import sys
import time

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path[:0] = [os.path.join(ROOT, "stark-perpetual_amd"), os.path.join(ROOT, "tests")]
import numpy as np  # noqa: E402
import torch  # noqa: E402
from starkperp import _lib, batch, batch_np as bn, state, stark  # noqa: E402
import workloads as wl  # noqa: E402


def c3_inputs():
    orders = wl.limit_orders(4096, seed=2)
    keys = wl.private_keys(1024, seed=12)
    from starkperp import perpetual_messages as pm
    zs = pm.limit_order_msgs_many([wl.order_args(o) for o in orders])
    pubs = batch.public_keys_many(keys)
    zsig = [z % 2**251 for z in zs]
    sigs = batch.sign_many(zsig, [keys[o["key_index"]] for o in orders])
    arr = {"sell": [], "buy": [], "fee": [], "a_sell": [], "a_buy": []}
    oa = [wl.order_args(o) for o in orders]
    for a in oa:
        syn, col, buying, f, a_syn, a_col, a_fee, nonce, pos, exp = a
        sd, bd, ns, nb = (col, syn, a_col, a_syn) if buying else (syn, col, a_syn, a_col)
        arr["sell"].append(sd); arr["buy"].append(bd); arr["fee"].append(f)
        arr["a_sell"].append(ns); arr["a_buy"].append(nb)
    u = lambda i: np.array([a[i] for a in oa], dtype=np.uint64)  # noqa: E731
    np_args = (bn.felts_from_ints(arr["sell"]), bn.felts_from_ints(arr["buy"]), bn.felts_from_ints(arr["fee"]),
               np.array(arr["a_sell"], dtype=np.uint64), np.array(arr["a_buy"], dtype=np.uint64), u(6), u(7), u(8), u(9))
    r_np, s_np = bn.felts_from_ints([r for r, _ in sigs]), bn.felts_from_ints([s for _, s in sigs])
    q_np = bn.felts_from_ints([pubs[o["key_index"]][0] for o in orders])
    amounts = bn.pack_fields(4096, [(np.array([o["amount_synthetic"] for o in orders], dtype=np.uint64), 0)])
    rng = random.Random(77)
    second = {rng.randrange(2**64): rng.randrange(1, 2**64) for _ in range(4096)}
    tree = state.LibrarySparseTree(64, 0)
    tree.update(second)  # existing state
    return np_args, r_np, s_np, q_np, amounts, tree


def one_call(inp):
    np_args, r_np, s_np, q_np, amounts, tree = inp
    t0 = time.perf_counter()
    w = bn.limit_order_words(*np_args)
    t1 = time.perf_counter()
    z, v, o, n, ok = bn.order_batch(w, r_np, s_np, q_np, tree, amounts)
    t2 = time.perf_counter()
    assert bool(ok) and bool((v == 1).all())
    return t2 - t0, t1 - t0


def main():
    mode = sys.argv[1] if len(sys.argv) > 1 else "calls"
    _lib.ensure_init()
    inp = c3_inputs()
    if mode == "calls":
        n = int(sys.argv[2]) if len(sys.argv) > 2 else 24
        ts = [one_call(inp) for _ in range(n)]
        tot = [t[0] for t in ts]
        steady = sorted(tot[2:])
        print("sp_order_batch, 4096 orders / 1024 keys, tree on existing state: %d calls (word packing in NumPy included: "
              "%.3f ms of each)" % (n, 1e3 * sum(t[1] for t in ts) / n))
        print("ms per call: " + " ".join("%.3f" % (1e3 * t) for t in tot))
        med = steady[len(steady) // 2]
        p90 = steady[min(len(steady) - 1, int(0.9 * len(steady)))]
        print("calls 3..%d: median %.3f ms  p90 %.3f ms (%.1f %% above the median)  min %.3f  max %.3f" % (
            n, 1e3 * med, 1e3 * p90, 100 * (p90 / med - 1), 1e3 * steady[0], 1e3 * steady[-1]))
    elif mode == "one":
        for _ in range(4):
            one_call(inp)
        torch.cuda.synchronize()
        time.sleep(0.3)
        t, _ = one_call(inp)
        torch.cuda.synchronize()
        print("the traced call: %.3f ms host-inclusive" % (1e3 * t))
    elif mode == "pmc":
        lib = _lib.load()
        dev = "cuda"
        stream = torch.cuda.current_stream().cuda_stream
        nv = 1 << 16
        rng = random.Random(21)
        dsk = [rng.randrange(1, batch.EC_ORDER) for _ in range(nv)]
        zv = [rng.randrange(2**251) for _ in range(nv)]
        kv = [rng.randrange(1, batch.EC_ORDER) for _ in range(nv)]
        pv = batch.public_keys_many(dsk)
        rv, sv, stv = batch.sign_attempt_many(zv, dsk, kv)
        dz, dr, dsig, dq = (stark.felts_to_tensor(v, dev) for v in (zv, rv, sv, [q[0] for q in pv]))
        res = torch.zeros(nv, dtype=torch.uint8, device=dev)
        for _ in range(2):
            _lib.check(lib.sp_ecdsa_verify_batch_dev(dz.data_ptr(), dr.data_ptr(), dsig.data_ptr(), dq.data_ptr(), None,
                                                     res.data_ptr(), nv, stream), "verify")
        batch.key_cache_reset()
        dslots = torch.from_numpy(np.asarray(batch.register_keys([q[0] for q in pv]), dtype=np.uint32).view(np.int32)).to(dev)
        for _ in range(2):
            _lib.check(lib.sp_ecdsa_verify_keyed_dev(dz.data_ptr(), dr.data_ptr(), dsig.data_ptr(), dslots.data_ptr(),
                                                     res.data_ptr(), nv, stream), "keyed")
        torch.cuda.synchronize()
        for _ in range(3):
            one_call(inp)
        torch.cuda.synchronize()
        print("pmc workload done: 2 x ladder 2^16, 2 x keyed 2^16, 3 order batches of 4096")
    inp[-1].close()


if __name__ == "__main__":
    main()

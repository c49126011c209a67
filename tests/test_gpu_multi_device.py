"""sp_init_devices (include/starkperp.h): several contexts in ONE process.  The GPU box has one device, so the two
contexts of this test sit on the same GPU ("a device may be listed twice") - every code path of the multi-context
library runs: lanes handed out round-robin over the contexts, per-context tables and scratch, context lookup by
device pointer for the _dev calls, the stateful entry points on the primary.  Runs in a fresh process (the test
session's own library is initialised with one context)."""
import os
import subprocess
import sys
import textwrap

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = textwrap.dedent(
    """
    import ctypes, random, sys, threading
    sys.path.insert(0, %(root)r)
    sys.path.insert(0, %(root)r + "/stark-perpetual_amd")
    import torch
    from oracle import cref
    from oracle import ref_py as R
    from starkperp import _lib, batch, stark, state

    lib = _lib.ensure_init()
    assert lib.sp_device_count() == 2, lib.sp_device_count()
    assert [_lib.context_info(i)[0] for i in range(2)] == [0, 0]
    assert lib.sp_init_devices(2, (ctypes.c_int * 2)(0, 0), 16) == 0          # same layout: idempotent
    assert lib.sp_init_devices(1, (ctypes.c_int * 1)(0), 16) != 0             # another layout: refused
    assert lib.sp_init(0, 16) == 0                                             # sp_init after it: no-op
    P, N = batch.FIELD_PRIME, batch.EC_ORDER
    batch.set_verify_policy(batch.VERIFY_POLICY_LADDER)

    def job(seed, out):
        rng = random.Random(seed)
        try:
            for it in range(4):
                n = rng.choice([1, 5, 300, 5000])
                xs = [rng.randrange(P) for _ in range(n)]
                ys = [rng.randrange(P) for _ in range(n)]
                assert batch.pedersen_hash_many(xs, ys) == cref.pedersen_hash_many(xs, ys)[0], "hash"
                m = rng.choice([1, 9])
                ds = [rng.randrange(1, N) for _ in range(m)]
                zs = [rng.randrange(2**251) for _ in range(m)]
                pubs = batch.public_keys_many(ds)
                assert pubs == cref.public_keys_many(ds), "public key"
                sigs = batch.sign_many(zs, ds)
                assert sigs[0] == R.sign(zs[0], ds[0]), "sign"
                assert batch.verify_codes(zs, [a for a, _ in sigs], [b for _, b in sigs], [q[0] for q in pubs]) == [1] * m
            out.append(None)
        except BaseException as e:
            out.append(e)

    results = []
    threads = [threading.Thread(target=job, args=(900 + i, results)) for i in range(8)]
    [t.start() for t in threads]
    [t.join() for t in threads]
    for r in results:
        if r is not None:
            raise r
    calls = [_lib.context_info(i)[1] for i in range(2)]
    assert min(calls) > 0, calls                                               # both contexts served host calls
    # _dev entry points: the context is found from the pointers' device
    rng = random.Random(5)
    xs = [rng.randrange(P) for _ in range(3000)]
    ys = [rng.randrange(P) for _ in range(3000)]
    dx, dy = stark.felts_to_tensor(xs), stark.felts_to_tensor(ys)
    out = torch.empty_like(dx)
    st = torch.cuda.current_stream().cuda_stream
    _lib.check(lib.sp_pedersen_batch_dev(dx.data_ptr(), dy.data_ptr(), out.data_ptr(), None, 3000, st), "dev")
    torch.cuda.synchronize()
    assert stark.tensor_to_felts(out) == cref.pedersen_hash_many(xs, ys)[0]
    # one large host batch: sliced over the contexts (one host thread each), results in place
    before = [_lib.context_info(i)[1] for i in range(2)]
    big = 40000
    bx = [rng.randrange(P) for _ in range(big)]
    by = [rng.randrange(P) for _ in range(big)]
    got = batch.pedersen_hash_many(bx, by)
    assert got[:64] == cref.pedersen_hash_many(bx[:64], by[:64])[0]
    assert got[-64:] == cref.pedersen_hash_many(bx[-64:], by[-64:])[0]
    assert got[big // 2 - 32 : big // 2 + 32] == cref.pedersen_hash_many(bx[big // 2 - 32 : big // 2 + 32], by[big // 2 - 32 : big // 2 + 32])[0]
    after = [_lib.context_info(i)[1] for i in range(2)]
    assert [a - b for a, b in zip(after, before)] == [1, 1], (before, after)
    m = 20000
    dsv = [rng.randrange(1, N) for _ in range(4)]
    pk = batch.public_keys_many(dsv)
    zs = [rng.randrange(2**251) for _ in range(m)]
    sg = batch.sign_many(zs, [dsv[i %% 4] for i in range(m)])
    zs[7] ^= 1
    zs[m - 3] ^= 1
    codes = batch.verify_codes(zs, [a for a, _ in sg], [b for _, b in sg], [pk[i %% 4][0] for i in range(m)])
    assert codes.count(0) == 2 and codes[7] == 0 and codes[m - 3] == 0
    leaves = [rng.randrange(P) for _ in range(1 << 9)]
    assert batch.merkle_root(leaves) == cref.merkle_levels(leaves)[-1][0]
    # stateful path (primary context): a persistent tree
    t = state.LibrarySparseTree(16, 0)
    old, new = t.update({3: 5, 77: 9})
    assert new == R.merkle_multi_update_sparse(16, {3: 5, 77: 9}, 0)
    t.close()
    lib.sp_shutdown()
    assert lib.sp_device_count() == 0
    assert lib.sp_init(0, 16) == 0 and lib.sp_device_count() == 1
    assert batch.pedersen_hash_many([1], [2]) == [R.pedersen_hash(1, 2)]
    print("multi-device ok", calls)
    """
)


def test_two_contexts_in_one_process():
    env = dict(os.environ, STARKPERP_DEVICES="0,0", STARKPERP_WINDOW_BITS="16")
    env.pop("LOCAL_RANK", None)
    out = subprocess.run([sys.executable, "-c", SCRIPT % {"root": ROOT}], capture_output=True, text=True, timeout=900,
                         env=env, cwd=ROOT)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
    assert "multi-device ok" in out.stdout

"""The C ABI from plain C: compiles tests/cabi/cabi_smoke.c with gcc against
include/starkperp.h + libstarkperp.so and runs it on the GPU."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_c_consumer_runs():
    libdir = os.path.join(ROOT, "stark-perpetual_amd", "lib")
    exe = os.path.join(ROOT, "tests", "cabi", "cabi_smoke")
    subprocess.check_call(["gcc", "-O1", os.path.join(ROOT, "tests", "cabi", "cabi_smoke.c"), "-o", exe,
                           "-L" + libdir, "-lstarkperp", "-Wl,-rpath," + libdir])
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 0, (out.returncode, out.stdout, out.stderr)
    assert "cabi_smoke ok" in out.stdout


@pytest.mark.parametrize("contexts", [1, 2])
def test_native_threads_program(contexts):
    """tests/cabi/cabi_threads.cpp - eight NATIVE host threads (no GIL anywhere) on every stateful entry point: hash
    batches / chains / rebuilds, AUTO and keyed verification on a 16-slot key cache with resets racing, persistent
    trees, sp_order_batch, both signers; every answer must equal the single-threaded one.  The same program is what
    tools/run_sanitizers.sh runs under ASan + UBSan and TSan (profiles/r05_sanitizers.txt)."""
    libdir = os.path.join(ROOT, "stark-perpetual_amd", "lib")
    exe = os.path.join(ROOT, "tests", "cabi", "cabi_threads")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-pthread", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cabi", "cabi_threads.cpp"), "-o", exe,
                           "-L" + libdir, "-lstarkperp", "-Wl,-rpath," + libdir])
    out = subprocess.run([exe, str(contexts), "2"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, (out.returncode, out.stdout[-1500:], out.stderr[-1500:])
    assert "cabi_threads ok" in out.stdout


def test_host_pointer_entry_points_from_concurrent_threads():
    """every batch call is re-entrant and thread-safe.  Eight host threads hammer the
    host-pointer entry points (which share one staging buffer inside the library; ctypes drops the
    GIL during the call) with different inputs; every result must equal the C oracle's."""
    import random
    import threading
    from oracle import cref
    from starkperp import batch

    def job(seed, out):
        rng = random.Random(seed)
        P = batch.FIELD_PRIME
        try:
            for it in range(6):
                n = rng.choice([1, 7, 300, 2000])
                xs = [rng.randrange(P) for _ in range(n)]
                ys = [rng.randrange(P) for _ in range(n)]
                assert batch.pedersen_hash_many(xs, ys) == cref.pedersen_hash_many(xs, ys)[0], (seed, it, "batch")
                leaves = [rng.randrange(P) for _ in range(1 << rng.choice([1, 4, 9]))]
                assert batch.merkle_root(leaves) == cref.merkle_levels(leaves)[-1][0], (seed, it, "root")
                words = [rng.randrange(P) for _ in range(rng.choice([2, 3, 5]))]
                acc = words[0]
                for w in words[1:]:
                    acc = cref.pedersen_hash_many([acc], [w])[0][0]
                assert batch.pedersen_chain(words) == acc, (seed, it, "chain")
            out.append(None)
        except BaseException as e:  # noqa: BLE001 - reported by the main thread
            out.append(e)

    results = []
    threads = [threading.Thread(target=job, args=(100 + i, results)) for i in range(8)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert len(results) == 8
    for r in results:
        if r is not None:
            raise r


def test_host_lanes_overlap_scalar_calls():
    """The lane entry points (hash / verify / sign / public key, include/starkperp.h "Threading") from
    twelve threads - more than the eight lanes - with different inputs: every result equals the oracle's, and
    one-item calls from eight threads finish in well under eight times the single-thread time."""
    import random
    import threading
    import time
    from oracle import cref
    from oracle import ref_py as R
    from starkperp import batch

    P, N = batch.FIELD_PRIME, batch.EC_ORDER
    batch.set_verify_policy(batch.VERIFY_POLICY_LADDER)
    try:
        def job(seed, out):
            rng = random.Random(seed)
            try:
                for it in range(4):
                    n = rng.choice([1, 3, 64, 700])
                    xs = [rng.randrange(P) for _ in range(n)]
                    ys = [rng.randrange(P) for _ in range(n)]
                    assert batch.pedersen_hash_many(xs, ys) == cref.pedersen_hash_many(xs, ys)[0], (seed, it, "hash")
                    m = rng.choice([1, 2, 33])
                    ds = [rng.randrange(1, N) for _ in range(m)]
                    zs = [rng.randrange(2**251) for _ in range(m)]
                    pubs = batch.public_keys_many(ds)
                    assert pubs == cref.public_keys_many(ds), (seed, it, "public key")
                    sigs = batch.sign_many(zs, ds)
                    assert sigs[0] == R.sign(zs[0], ds[0]), (seed, it, "sign")
                    rs, ss = [a for a, _ in sigs], [b for _, b in sigs]
                    if m > 1:
                        zs[1] = (zs[1] + 1) % 2**251
                    want = [1] * m
                    if m > 1:
                        want[1] = 0
                    assert batch.verify_codes(zs, rs, ss, [q[0] for q in pubs]) == want, (seed, it, "verify")
                out.append(None)
            except BaseException as e:  # noqa: BLE001 - reported by the main thread
                out.append(e)

        results = []
        threads = [threading.Thread(target=job, args=(500 + i, results)) for i in range(12)]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        assert len(results) == 12
        for r in results:
            if r is not None:
                raise r

        # overlap: 8 threads x 40 one-signature verifications against 40 from one thread
        d = 4242
        q = R.private_key_to_ec_point_on_stark_curve(d)
        z = 0x1234567
        r, s = R.sign(z, d)

        def spin(count):
            for _ in range(count):
                assert batch.verify_codes([z], [r], [s], [q[0]]) == [1]

        spin(5)
        t0 = time.perf_counter()
        spin(40)
        one = time.perf_counter() - t0
        threads = [threading.Thread(target=spin, args=(40,)) for _ in range(8)]
        t0 = time.perf_counter()
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        eight = time.perf_counter() - t0
        print("one thread: %.2f ms per call; eight threads: %.2f ms per call (aggregate)" % (
            one / 40 * 1e3, eight / 320 * 1e3))
        assert eight < 6 * one, (one, eight)  # serialised calls would take 8 x (measured: 1 - 2 x; the margin is for a busy host)
    finally:
        batch.set_verify_policy(batch.VERIFY_POLICY_AUTO)

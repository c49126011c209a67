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

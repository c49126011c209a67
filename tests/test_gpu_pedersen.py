"""GPU parity: batched Pedersen / Merkle through the C ABI vs the golden vectors produced by the
reference and vs the oracle on seeded inputs."""
import json
import os
import random

import pytest

import workloads as wl
from oracle import ref_py as R

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
P = R.FIELD_PRIME


def load(name):
    return json.load(open(os.path.join(GOLD, name)))


def h(s):
    return int(s, 16)


@pytest.fixture(scope="module")
def batch():
    from starkperp import batch as b
    return b


def test_g1_full_batch(batch):
    g = load("g1_pedersen.json")
    pairs = wl.pedersen_pairs(g["n"], seed=g["seed"])
    out = batch.pedersen_hash_many([p[0] for p in pairs], [p[1] for p in pairs])
    assert out == [h(v) for v in g["all"]]
    assert wl.digest_felts(out) == g["digest"]


def test_edges_and_kats(batch):
    g = load("g1_pedersen.json")
    xs = [h(a) for a, _, _ in g["edge"]]
    ys = [h(b) for _, b, _ in g["edge"]]
    assert batch.pedersen_hash_many(xs, ys) == [h(o) for _, _, o in g["edge"]]
    k = load("reference_kats.json")
    for case in k["hash_test"].values():
        assert batch.pedersen_hash_many([h(case["input_1"])], [h(case["input_2"])]) == [h(case["output"])]
    # ragged sizes around wave / block boundaries
    rng = random.Random(5)
    for n in (1, 2, 63, 64, 65, 255, 257):
        xs = [rng.randrange(P) for _ in range(n)]
        ys = [rng.randrange(P) for _ in range(n)]
        got = batch.pedersen_hash_many(xs, ys)
        idx = sorted(set([0, n - 1, n // 2]))
        for i in idx:
            assert got[i] == R.pedersen_hash(xs[i], ys[i])
    assert batch.pedersen_hash_many([], []) == []


def test_out_of_range_raises(batch):
    with pytest.raises(AssertionError):
        batch.pedersen_hash_many([P], [0])


def test_chain(batch):
    rng = random.Random(9)
    for n in (1, 2, 5):
        el = [rng.randrange(P) for _ in range(n)]
        exp = el[0]
        for e in el[1:]:
            exp = R.pedersen_hash(exp, e)
        assert batch.pedersen_chain(el) == exp


def test_chains_in_one_launch_vs_c_oracle(batch):
    """Batches of equal-depth chains in every size class of the fused chain kernel (<= 2048 / 4096 / 8192 chains: eight,
    four, two quads per hash) and just beyond it (one launch per step), depths 2 .. 9, against the C oracle step by
    step; a single long chain folded from the left and from the right (the program-hash shape); an out-of-range
    word in the middle of a chain is reported, not reduced."""
    from oracle import cref
    from starkperp import _lib, hash_chains as hc
    rng = random.Random(123)
    for width, depth in ((1, 9), (7, 3), (300, 5), (2048, 4), (2049, 3), (4096, 4), (5000, 3), (8192, 3), (8193, 3),
                         (64, 2)):
        chains = [[rng.randrange(P) for _ in range(depth)] for _ in range(width)]
        exp = [c[0] for c in chains]
        for j in range(1, depth):
            exp, st = cref.pedersen_hash_many(exp, [c[j] for c in chains])
            assert not any(st)
        assert batch.pedersen_chains_many(chains) == exp, (width, depth)
    words = [rng.randrange(P) for _ in range(200)]
    left = words[0]
    for w in words[1:]:
        left = cref.pedersen_hash_many([left], [w])[0][0]
    assert batch.pedersen_chain(words) == left
    right = words[-1]
    for w in reversed(words[:-1]):
        right = cref.pedersen_hash_many([w], [right])[0][0]
    assert hc.compute_hash_chain(words) == right
    # status flag of the fused launch: word 2 of chain 5 is p itself
    lib = _lib.ensure_init()
    width, depth = 16, 4
    flat = [rng.randrange(P) for _ in range(width * depth)]
    flat[2 * width + 5] = P
    out, st = _lib.new_felts(width), _lib.new_bytes(1)
    _lib.check(lib.sp_pedersen_chains(_lib.pack_felts(flat), width, depth, out, st), "sp_pedersen_chains")
    assert st[0] & 1  # SP_HASH_OUT_OF_RANGE


def test_merkle_small(batch):
    g = load("g6_merkle.json")
    for hgt in range(0, 11):
        lv = wl.leaves(1 << hgt, seed=100 + hgt)
        assert batch.merkle_root(lv) == h(g["roots_seed_100_plus_h"][str(hgt)])
    lv = wl.leaves(64, seed=106)
    assert batch.merkle_levels(lv) == R.merkle_levels(lv)


def test_merkle_c2_full(batch):
    g = load("g6_c2_tree.json")
    lv = wl.leaves(1 << 16, seed=g["seed"])
    levels = batch.merkle_levels(lv)
    assert levels[-1][0] == h(g["root"])
    assert [hex(l[0]) for l in levels] == g["left_spine"]
    assert [wl.digest_felts(l) for l in levels] == g["level_digests"]


def test_shutdown_and_reinit_with_other_window(batch):
    """sp_shutdown releases every device buffer; a re-init with a different window width gives the
    same hashes (the tables are an implementation detail)."""
    from starkperp import _lib
    lib = _lib.load()
    g = load("g1_pedersen.json")
    pairs = wl.pedersen_pairs(64, seed=g["seed"])
    exp = [h(v) for v in g["all"][:64]]
    for wbits in (8, 13, 16):
        lib.sp_shutdown()
        assert lib.sp_is_initialised() == 0
        _lib.check(lib.sp_init(0, wbits), "sp_init")
        assert lib.sp_window_bits() == wbits
        assert batch.pedersen_hash_many([p[0] for p in pairs], [p[1] for p in pairs]) == exp
        lv = wl.leaves(256, seed=108)
        assert batch.merkle_root(lv) == h(load("g6_merkle.json")["roots_seed_100_plus_h"]["8"])
    lib.sp_shutdown()
    _lib.ensure_init()


def test_table_build_schedules_agree(batch):
    """Round 4: the tables are built by doubling passes (one affine addition per entry).  The round 1 - 3 build (every
    entry from its set bits, STARKPERP_TABLE_BUILD=direct) in a subprocess must hash the reference's goldens and 4096
    random pairs exactly like this process does - under a window width whose windows take several doubling passes."""
    import subprocess
    import sys
    from starkperp import _lib
    g = load("g1_pedersen.json")
    lib = _lib.load()
    lib.sp_shutdown()
    _lib.check(lib.sp_init(0, 18), "sp_init")  # 2^18-entry windows: five doubling passes on top of the 2^13 seed entries
    pairs = wl.pedersen_pairs(g["n"], seed=g["seed"])
    rng = random.Random(5)
    extra = [(rng.randrange(P), rng.randrange(P)) for _ in range(4096)]
    ours = batch.pedersen_hash_many([p[0] for p in pairs + extra], [p[1] for p in pairs + extra])
    assert ours[: g["n"]] == [h(v) for v in g["all"]]
    lib.sp_shutdown()
    _lib.ensure_init()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = r'''
import sys, random, hashlib
sys.path[:0] = [%r, %r, %r]
from starkperp import batch
import workloads as wl
from oracle import ref_py as R
pairs = wl.pedersen_pairs(%d, seed=%d)
rng = random.Random(5)
extra = [(rng.randrange(R.FIELD_PRIME), rng.randrange(R.FIELD_PRIME)) for _ in range(4096)]
out = batch.pedersen_hash_many([p[0] for p in pairs + extra], [p[1] for p in pairs + extra])
print(hashlib.sha256(repr(out).encode()).hexdigest())
''' % (root, os.path.join(root, "stark-perpetual_amd"), os.path.join(root, "tests"), g["n"], g["seed"])
    env = dict(os.environ, STARKPERP_TABLE_BUILD="direct", STARKPERP_WINDOW_BITS="18")
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0, out.stderr[-1500:]
    import hashlib
    assert out.stdout.split()[-1] == hashlib.sha256(repr(ours).encode()).hexdigest()


def test_bulk_and_split_paths_vs_c_oracle(batch):
    """Every accumulate variant (1, 2, 4, 8 lanes per hash are picked by batch size) against the C
    oracle on seeded inputs, plus out-of-range status propagation inside a large batch."""
    from oracle import cref
    rng = random.Random(77)
    for n in (70000, 40000, 20000, 3000):
        xs = [rng.randrange(P) for _ in range(n)]
        ys = [rng.randrange(P) for _ in range(n)]
        exp, st = cref.pedersen_hash_many(xs, ys)
        assert not any(st)
        assert batch.pedersen_hash_many(xs, ys) == exp, n


def test_forest_roots_equal_individual_roots(batch):
    trees = [wl.leaves(256, seed=300 + i) for i in range(7)]
    assert batch.merkle_roots_many(trees) == [batch.merkle_root(t) for t in trees]
    g = load("g6_merkle.json")
    assert batch.merkle_roots_many([wl.leaves(1 << 10, seed=110)]) == [h(g["roots_seed_100_plus_h"]["10"])]
    assert batch.merkle_roots_many([[5], [7]]) == [5, 7]


def test_every_pair_of_extreme_limb_patterns_vs_c_oracle(batch):
    """All ordered pairs of the extreme-limb-pattern felts (tests/workloads.py extreme_felts: all-ones limbs,
    p - small, powers of two at the limb boundaries, ...) through the bulk path and, in small slices, through
    every latency kernel, against the optimised C comparator (itself pinned by the reference goldens)."""
    from oracle import cref
    ext = wl.extreme_felts()
    xs = [a for a in ext for _ in ext]
    ys = [b for _ in ext for b in ext]
    exp, st = cref.opt_pedersen_hash_many(xs, ys)
    assert not any(st)
    assert batch.pedersen_hash_many(xs, ys) == exp
    for size in (1, 3, 17, 64, 200, 1000, 3000):  # quad / split / fused kernels by level size
        for off in range(0, min(len(xs), 6 * size), size):
            assert batch.pedersen_hash_many(xs[off:off + size], ys[off:off + size]) == exp[off:off + size], (size, off)

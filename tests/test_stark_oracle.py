"""Algebraic self-checks of the prover-side oracle (oracle/stark_ref.py).  The reference has no
prover, so these replace golden vectors for rows A10-A13 ("parity unpinned")."""
import random

import pytest

from oracle import ref_py as R
from oracle import stark_ref as S

P = S.P


def test_roots_of_unity():
    w = S.root_of_unity(20)
    assert w == 0x0594BEAFCA8A00D9581D81CAEE93DC85C727C9AF7FC4C648E3D47B998574E81F  # SURVEY appx B
    assert pow(w, 1 << 20, P) == 1 and pow(w, 1 << 19, P) == P - 1


def test_ntt_matches_naive_and_inverts():
    rng = random.Random(1)
    for log_n in (0, 1, 3, 6):
        n = 1 << log_n
        c = [rng.randrange(P) for _ in range(n)]
        w = S.root_of_unity(log_n)
        ev = S.ntt(c, w)
        assert ev == S.naive_dft(c, w)
        assert S.intt(ev, w) == c


def test_lde_agrees_with_polynomial():
    rng = random.Random(2)
    n = 16
    coeffs = [rng.randrange(P) for _ in range(n)]
    evals = S.ntt(coeffs, S.root_of_unity(4))
    ext = S.lde(evals)
    w = S.root_of_unity(6)
    for i in (0, 1, 17, 63):
        x = S.GEN * pow(w, i, P) % P
        assert ext[i] == sum(c * pow(x, k, P) for k, c in enumerate(coeffs)) % P


def test_trace_satisfies_constraints_and_composition_is_low_degree():
    rng = random.Random(3)
    inputs = [(rng.randrange(P), rng.randrange(P))]
    cols = S.pedersen_trace(inputs)
    n = len(cols[0])
    assert n == 512
    # last accumulator row carries the hash
    assert cols[1][511] == R.pedersen_hash(*inputs[0])
    per = S.periodic_columns()
    for i in range(n):
        cur = [c[i] for c in cols]
        nxt = [c[(i + 1) % n] for c in cols]
        assert all(v == 0 for v in S.constraint_values(cur, nxt, [t[i] for t in per])), i
    trace_lde = [S.lde(c) for c in cols]
    per_lde = S.periodic_lde(n)
    alphas = [rng.randrange(P) for _ in range(S.N_CONSTRAINTS)]
    comp = S.composition_on_coset(trace_lde, per_lde, n, alphas)
    assert S.poly_degree_bound_check(comp, S.GEN, 3 * n - 1)
    # a corrupted trace does not give a low-degree quotient
    bad = [list(c) for c in cols]
    bad[1][100] = (bad[1][100] + 1) % P
    comp_bad = S.composition_on_coset([S.lde(c) for c in bad], per_lde, n, alphas)
    assert not S.poly_degree_bound_check(comp_bad, S.GEN, 3 * n - 1)
    # FRI: folding halves the degree bound; fold equals direct evaluation of the folded polynomial
    layer, shift, bound = comp, S.GEN, 3 * n
    while len(layer) > 64:
        beta = rng.randrange(P)
        layer = S.fri_fold(layer, beta, shift)
        shift = shift * shift % P
        bound //= 2
        assert S.poly_degree_bound_check(layer, shift, bound - 1)
    assert len(layer) == 64 and bound == 48


def test_fold_matches_even_odd_split():
    rng = random.Random(4)
    m = 32
    coeffs = [rng.randrange(P) for _ in range(m)]
    shift = 5
    w = S.root_of_unity(5)
    vals = [sum(c * pow(shift * pow(w, i, P) % P, k, P) for k, c in enumerate(coeffs)) % P
            for i in range(m)]
    beta = rng.randrange(P)
    folded = S.fri_fold(vals, beta, shift)
    g = [(coeffs[2 * k] + beta * coeffs[2 * k + 1]) % P for k in range(m // 2)]
    w2 = w * w % P
    for i in range(m // 2):
        y = shift * shift * pow(w2, i, P) % P
        assert folded[i] == sum(c * pow(y, k, P) for k, c in enumerate(g)) % P


def test_commit_rows_small():
    cols = [[1, 2, 3, 4], [5, 6, 7, 8]]
    leaves = [R.pedersen_hash(a, b) for a, b in zip(*cols)]
    assert S.commit_rows(cols) == R.merkle_root(leaves)
    assert S.commit_rows([[1, 2, 3, 4]]) == R.merkle_root([1, 2, 3, 4])


def test_ec_ladder_air_trace_and_degree():
    rng = random.Random(8)
    q = R.ec_mult(rng.randrange(1, R.EC_ORDER), tuple(R.EC_GEN))
    m = rng.randrange(1, 2**251)
    inputs = [(m, q), (2**251 - 1, tuple(R.EC_GEN))]
    cols = S.ec_ladder_trace(inputs)
    n = len(cols[0])
    assert n == 512
    assert (cols[1][251], cols[2][251]) == R.mimic_ec_mult_air(m, q, R.SHIFT_POINT)
    assert (cols[1][256 + 251], cols[2][256 + 251]) == R.mimic_ec_mult_air(2**251 - 1, R.EC_GEN, R.SHIFT_POINT)
    per = S.ec_ladder_periodic_columns()
    for i in range(n):
        cur = [c[i] for c in cols]
        nxt = [c[(i + 1) % n] for c in cols]
        vals = S.ec_ladder_constraint_values(cur, nxt, [t[i % 256] for t in per])
        assert all(v == 0 for v in vals), (i, vals)
    alphas = [rng.randrange(P) for _ in range(S.N_EC_LADDER_CONSTRAINTS)]
    per_lde = S.periodic_lde(n, air="ec_ladder")
    comp = S.composition_on_coset([S.lde(c) for c in cols], per_lde, n, alphas, air="ec_ladder")
    assert S.poly_degree_bound_check(comp, S.GEN, 3 * n - 1)
    bad = [list(c) for c in cols]
    bad[6][10] = (bad[6][10] + 1) % P
    comp_bad = S.composition_on_coset([S.lde(c) for c in bad], per_lde, n, alphas, air="ec_ladder")
    assert not S.poly_degree_bound_check(comp_bad, S.GEN, 3 * n - 1)


def test_range_check_air_trace_and_degree():
    """0 <= value < 2^128 by bit decomposition: constraints vanish on honest traces (edge values included), the
    composition has degree < 3n, and a value of 2^128 (or a flipped cell) breaks it."""
    rng = random.Random(12)
    values = [0, 1, 2**128 - 1, 2**64, rng.randrange(2**128), rng.randrange(2**64), rng.randrange(2**32), 2**127]
    cols = S.range_check_trace(values)
    n = len(cols[0])
    assert n == 1024 and [cols[0][128 * k] for k in range(8)] == values
    per = S.range_check_periodic_columns()
    for i in range(n):
        vals = S.range_check_constraint_values([cols[0][i]], [cols[0][(i + 1) % n]], [t[i % 128] for t in per])
        assert vals == [0, 0], (i, vals)
    alphas = [rng.randrange(P) for _ in range(S.N_RANGE_CHECK_CONSTRAINTS)]
    per_lde = S.periodic_lde(n, air="range_check")
    comp = S.composition_on_coset([S.lde(c) for c in cols], per_lde, n, alphas, air="range_check")
    assert S.poly_degree_bound_check(comp, S.GEN, 3 * n - 1)
    for bad_values in ([2**128] + values[1:], values[:3] + [2**200] + values[4:]):
        bad = S.range_check_trace(bad_values)
        comp_bad = S.composition_on_coset([S.lde(c) for c in bad], per_lde, n, alphas, air="range_check")
        assert not S.poly_degree_bound_check(comp_bad, S.GEN, 3 * n - 1)
    flipped = [list(cols[0])]
    flipped[0][300] ^= 2
    comp_bad = S.composition_on_coset([S.lde(c) for c in flipped], per_lde, n, alphas, air="range_check")
    assert not S.poly_degree_bound_check(comp_bad, S.GEN, 3 * n - 1)


def test_ecdsa_air_trace_and_degree():
    """The ECDSA-verification AIR (three linked ladders, signature.py:217-260): every constraint vanishes
    on the witness of signatures the reference accepts, the composition has degree < 3n, and tampering with
    the link between the ladders (B), with r, or with a carry is caught."""
    rng = random.Random(18)
    insts = []
    for _ in range(2):
        d = rng.randrange(1, R.EC_ORDER)
        z = rng.randrange(1, 2**251)
        r, s = R.sign(z, d)
        insts.append(S.ecdsa_instance(z, r, s, R.private_key_to_ec_point_on_stark_curve(d)))
    cols = S.ecdsa_trace(insts)
    n = len(cols[0])
    assert n == 2048 and len(cols) == 10
    per = S.ecdsa_periodic_columns()
    for i in range(n):
        vals = S.ecdsa_constraint_values([c[i] for c in cols], [c[(i + 1) % n] for c in cols], [t[i % 1024] for t in per])
        assert all(v == 0 for v in vals), (i, [k for k, v in enumerate(vals) if v])
    # the third ladder's base point is zG + rQ, its output minus the shift point has x = r
    z, r, w, q = insts[0]
    b = R.ec_add(R.mimic_ec_mult_air(z, R.EC_GEN, R.MINUS_SHIFT_POINT), R.mimic_ec_mult_air(r, q, R.SHIFT_POINT))
    assert (cols[3][512], cols[4][512]) == b
    assert R.ec_add((cols[1][767], cols[2][767]), R.MINUS_SHIFT_POINT)[0] == r
    alphas = [rng.randrange(P) for _ in range(S.N_ECDSA_CONSTRAINTS)]
    per_lde = S.periodic_lde(n, air="ecdsa")
    comp = S.composition_on_coset([S.lde(c) for c in cols], per_lde, n, alphas, air="ecdsa")
    assert S.poly_degree_bound_check(comp, S.GEN, 3 * n - 1)
    for col, row in ((3, 512), (9, 600), (7, 300), (0, 256)):
        bad = [list(c) for c in cols]
        bad[col][row] = (bad[col][row] + 1) % P
        comp_bad = S.composition_on_coset([S.lde(c) for c in bad], per_lde, n, alphas, air="ecdsa")
        assert not S.poly_degree_bound_check(comp_bad, S.GEN, 3 * n - 1), (col, row)
    # a signature the reference rejects has no instance
    import pytest
    with pytest.raises(AssertionError):
        S.ecdsa_instance(z + 1, insts[0][1], R.sign(z, 5)[1], q)


def test_rc16_builtin_air_oracle():
    """The range-check builtin's encoding: the composition of a valid two-phase trace is a
    polynomial of degree < 2n, a corrupted limb / a sorted column that is no permutation / a wrong limb range
    are not; the padding fills the holes of the limb range."""
    import random
    rng = random.Random(5)
    vals = [sum(rng.randrange(100, 140) << (16 * k) for k in range(8)) for _ in range(12)]
    padded, lo, hi = S.rc16_fill(vals, 16)
    assert padded[:12] == vals and (lo, hi) == (100, 139)
    cols = S.rc16_trace(padded)
    n = len(cols[0])
    assert n == 128 and sorted(set(cols[0])) == list(range(lo, hi + 1))
    assert [cols[1][8 * v + 7] for v in range(16)] == padded
    z = rng.randrange(S.P)
    p = S.rc16_product_column(cols[0], cols[2], z)
    assert p[-1] == 1
    alphas = [rng.randrange(S.P) for _ in range(S.N_RC16_CONSTRAINTS)]

    def low_degree(c, pc, rc_lo=lo, rc_hi=hi):
        comp = S.rc16_composition_on_coset([S.lde(x) for x in c], S.lde(pc), n, alphas, z, rc_lo, rc_hi)
        return S.poly_degree_bound_check(comp, S.GEN, 2 * n)
    assert low_degree(cols, p)
    bad = [list(c) for c in cols]
    bad[0][5] += 1                                   # a limb that is not in the pool any more
    assert not low_degree(bad, p)
    assert not low_degree(bad, S.rc16_product_column(bad[0], bad[2], z))   # ... even with its own product column
    bad = [list(c) for c in cols]
    bad[1][9] += 1                                   # the running value does not follow the limbs
    assert not low_degree(bad, p)
    assert not low_degree(cols, p, rc_lo=lo + 1) and not low_degree(cols, p, rc_hi=hi + 1)
    with pytest.raises(AssertionError):
        S.rc16_fill([1 << 128], 4)
    with pytest.raises(AssertionError):
        S.rc16_fill([0, 0xFFFF], 16)                 # 65 534 holes do not fit 16 values

"""Host-compiled (g++, bound checks on) unit tests of the device arithmetic headers
stark-perpetual_amd/csrc/{fp29,curve}.hpp against Python integers / the oracle."""
import ctypes
import os
import random
import subprocess

import pytest

from oracle import ref_py as R

HERE = os.path.dirname(os.path.abspath(__file__))
P, N = R.FIELD_PRIME, R.EC_ORDER


@pytest.fixture(scope="module")
def shim():
    so = os.path.join(HERE, "host", "host_shim.so")
    src = os.path.join(HERE, "host", "host_shim.cpp")
    subprocess.check_call(["g++", "-O2", "-shared", "-fPIC", "-o", so, src])
    return ctypes.CDLL(so)


def W(v):
    return (ctypes.c_uint32 * 8)(*[(v >> (32 * i)) & 0xFFFFFFFF for i in range(8)])


def WN(vals):
    arr = (ctypes.c_uint32 * (8 * len(vals)))()
    for k, v in enumerate(vals):
        for i in range(8):
            arr[8 * k + i] = (v >> (32 * i)) & 0xFFFFFFFF
    return arr


def I(buf, off=0):
    return sum(buf[off + i] << (32 * i) for i in range(8))


EDGE = [0, 1, 2, P - 1, P - 2, 2**251, 2**192, 2**29 - 1, 2**232, (P - 1) // 2]


def test_constants():
    Rm = 2**261
    limbs = lambda v: [(v >> (29 * i)) & (2**29 - 1) for i in range(9)]
    import re
    hdr = open(os.path.join(HERE, "..", "stark-perpetual_amd", "csrc", "fp29.hpp")).read()

    def grab(name):
        m = re.search(name + r" = \{\{([^}]*)\}\}", hdr)
        return [int(x, 16) for x in m.group(1).replace("\n", " ").split(",")]

    assert grab("FE_ONE_M") == limbs(Rm % P)
    assert grab("FE_R2") == limbs(Rm * Rm % P)
    assert grab("FN_ONE_M") == limbs(Rm % N)
    assert grab("FN_R2") == limbs(Rm * Rm % N)
    assert grab("FN_N") == limbs(N)
    assert limbs(P) == [1, 0, 0, 0, 0, 0, 0x440000, 0, 0x80000]
    assert (-pow(N, -1, 2**29)) % 2**29 == 0x8BDE631


def test_pack_roundtrip(shim):
    rng = random.Random(1)
    out = (ctypes.c_uint32 * 8)()
    for v in EDGE + [rng.randrange(2**256) for _ in range(200)]:
        shim.t_roundtrip(W(v), out)
        assert I(out) == v


def test_fe_mul_sqr_inv(shim):
    rng = random.Random(2)
    out = (ctypes.c_uint32 * 8)()
    vals = EDGE + [rng.randrange(P) for _ in range(300)]
    for a in vals:
        b = rng.choice(vals)
        shim.t_fe_mul(W(a), W(b), out)
        assert I(out) == a * b % P
        shim.t_fe_sqr(W(a), out)
        assert I(out) == a * a % P
    for a in vals[:40]:
        if a == 0:
            continue
        shim.t_fe_inv(W(a), out)
        assert I(out) == pow(a, -1, P)
        assert shim.t_fe_is_qr(W(a)) == (1 if pow(a, (P - 1) // 2, P) == 1 else 0)


def test_fe_inv_gcd(shim):
    rng = random.Random(7)
    out = (ctypes.c_uint32 * 8)()
    Rm = 2**261
    import re
    hdr = open(os.path.join(HERE, "..", "stark-perpetual_amd", "csrc", "fp29.hpp")).read()
    m = re.search(r"FE_R3 = \{\{([^}]*)\}\}", hdr)
    assert [int(x, 16) for x in m.group(1).replace("\n", " ").split(",")] == [
        (pow(Rm, 3, P) >> (29 * i)) & (2**29 - 1) for i in range(9)]
    vals = [1, 2, 3, P - 1, P - 2, 2**251, 2**192 + 1, (P + 1) // 2] + [rng.randrange(1, P) for _ in range(400)]
    for a in vals:
        shim.t_fe_inv_plain_gcd(W(a), out)
        assert I(out) == pow(a, -1, P), hex(a)
        shim.t_fe_inv_gcd(W(a), out)
        assert I(out) == pow(a, -1, P), hex(a)
    shim.t_fe_inv_plain_gcd(W(0), out)
    assert I(out) == 0


def test_fe_inv_gcd_variable_time(shim):
    """divsteps_29_var (ctz jumps + six-bit cancellation, delta = 1 start): the inversion every hash and
    verification goes through."""
    rng = random.Random(8)
    out = (ctypes.c_uint32 * 8)()
    vals = [1, 2, 3, 4, 5, P - 1, P - 2, 2**251, 2**192 + 1, (P + 1) // 2, 2**29, 2**29 - 1, 2**58 + 1, P - 2**29]
    vals += [2**k for k in range(0, 252, 7)] + [P - 2**k for k in range(1, 251, 11)]
    vals += [rng.randrange(1, P) for _ in range(3000)] + [rng.randrange(1, 2**64) for _ in range(200)]
    for a in vals:
        shim.t_fe_inv_plain_gcd_var(W(a), out)
        assert I(out) == pow(a, -1, P), hex(a)
    for a in vals[:300]:
        shim.t_fe_inv_gcd_var(W(a), out)
        assert I(out) == pow(a, -1, P), hex(a)
    shim.t_fe_inv_plain_gcd_var(W(0), out)
    assert I(out) == 0


def test_fe_lazy_expr(shim):
    rng = random.Random(3)
    out = (ctypes.c_uint32 * 8)()
    vals = EDGE + [rng.randrange(P) for _ in range(50)]
    for _ in range(500):
        a, b, c, d, e, f = (rng.choice(vals) for _ in range(6))
        shim.t_fe_expr(W(a), W(b), W(c), W(d), W(e), W(f), out)
        assert I(out) == ((a - b) * (c + d) - e * f) % P


def test_fn(shim):
    rng = random.Random(4)
    out = (ctypes.c_uint32 * 8)()
    vals = [0, 1, 2, N - 1, N - 2, 2**251] + [rng.randrange(N) for _ in range(200)]
    for a in vals:
        b = rng.choice(vals)
        shim.t_fn_mul(W(a), W(b), out)
        assert I(out) == a * b % N
    for a in vals[1:]:
        shim.t_fn_inv(W(a), out)
        assert I(out) == pow(a, -1, N)
    for a in vals[1:12]:
        shim.t_fn_inv_fermat(W(a), out)
        assert I(out) == pow(a, -1, N)
        shim.t_fe_inv_fermat(W(a), out)
        assert I(out) == pow(a, -1, P)


def rand_point(rng):
    return R.ec_mult(rng.randrange(1, N), tuple(R.EC_GEN))


def test_xyzz_chain_and_add(shim):
    rng = random.Random(5)
    x = (ctypes.c_uint32 * 8)()
    y = (ctypes.c_uint32 * 8)()
    for n in (2, 3, 5, 17):
        pts = [rand_point(rng) for _ in range(n)]
        exp = pts[0]
        for q in pts[1:]:
            exp = R.ec_add(exp, q)
        shim.t_xyzz_chain(WN([p[0] for p in pts]), WN([p[1] for p in pts]), n, x, y)
        assert (I(x), I(y)) == exp
    pts = [rand_point(rng) for _ in range(4)]
    exp = R.ec_add(R.ec_add(pts[0], pts[1]), R.ec_add(pts[2], pts[3]))
    shim.t_xyzz_add(WN([p[0] for p in pts]), WN([p[1] for p in pts]), x, y)
    assert (I(x), I(y)) == exp


def test_jacobian_full_add(shim):
    rng = random.Random(12)
    x = (ctypes.c_uint32 * 8)()
    y = (ctypes.c_uint32 * 8)()
    for _ in range(6):
        p, q = rand_point(rng), rand_point(rng)
        shim.t_jac_add(W(p[0]), W(p[1]), W(q[0]), W(q[1]), x, y)
        assert (I(x), I(y)) == R.ec_add(R.ec_mult(2, p), R.ec_mult(3, q))


def test_jacobian_ladder(shim):
    rng = random.Random(6)
    x = (ctypes.c_uint32 * 8)()
    y = (ctypes.c_uint32 * 8)()
    for k in [1, 2, 3, 5, N - 1, 2**251 - 1] + [rng.randrange(1, N) for _ in range(6)]:
        q = rand_point(rng)
        shim.t_jac_mul(W(q[0]), W(q[1]), W(k), x, y)
        assert (I(x), I(y)) == R.ec_mult(k, q)


def test_fe_half(shim):
    """fe_half (the 1/2 of the FRI fold without a multiplication): (a - b) / 2 mod p for a, b < p."""
    rng = random.Random(12)
    out = (ctypes.c_uint32 * 8)()
    inv2 = pow(2, -1, P)
    pairs = [(0, 0), (1, 0), (0, 1), (P - 1, 0), (0, P - 1), (P - 1, P - 2), (2**251, 1), (5, 2**200)]
    pairs += [(rng.randrange(P), rng.randrange(P)) for _ in range(2000)]
    for a, b in pairs:
        shim.t_fe_half(W(a), W(b), out)
        assert I(out) == (a - b) * inv2 % P, (hex(a), hex(b))


def test_fn_inv_variable_time(shim):
    """fn_inv_var: the mod-N inversion of verification (public scalars) in its variable-time form."""
    rng = random.Random(13)
    out = (ctypes.c_uint32 * 8)()
    vals = [1, 2, 3, N - 1, N - 2, 2**251, 2**250 + 1, (N + 1) // 2] + [rng.randrange(1, N) for _ in range(2000)]
    vals += [2**k for k in range(0, 251, 9)] + [N - 2**k for k in range(0, 250, 13)] + [N // k for k in range(2, 30)]
    for a in vals:
        shim.t_fn_inv_var(W(a), out)                 # double-steered form, divsteps fallback for the edge values
        assert I(out) == pow(a, -1, N), hex(a)
    for a in vals[:400]:
        shim.t_fn_inv_divsteps_var(W(a), out)        # the fallback on its own
        assert I(out) == pow(a, -1, N), hex(a)


def test_fe_canon_on_raw_limbs(shim):
    """fe_canon takes any N-form or lazy value (|value| < 16 p, limbs up to 2 * 2^29 in magnitude, signed) to
    [0, p): multiples of p, the boundaries around them, negative values, unnormalised limbs."""
    rng = random.Random(99)

    def limbs_of(v):  # normalised signed representation: limbs 0..7 in [0, 2^29), limb 8 carries the sign
        out = [(v >> (29 * i)) & (2**29 - 1) for i in range(8)]
        out.append(v >> 232)
        return out

    def check(limbs):
        value = sum(l << (29 * i) for i, l in enumerate(limbs))
        out = (ctypes.c_uint32 * 8)()
        shim.t_fe_canon_limbs((ctypes.c_int32 * 9)(*limbs), out)
        assert I(out) == value % P, (limbs, value)

    for k in range(-15, 16):
        for d in (-2, -1, 0, 1, 2, 2**29, -(2**29), 2**192 * 17, 2**250):
            check(limbs_of(k * P + d))
    for _ in range(2000):
        v = rng.randrange(-16 * P + 1, 16 * P)
        check(limbs_of(v))
        # the same value with lazy limbs: move one unit of limb i + 1 down into limb i, or borrow the other way
        l = limbs_of(v)
        i = rng.randrange(8)
        if rng.random() < 0.5:
            l[i] += 2**29
            l[i + 1] -= 1
        else:
            l[i] -= 2**29
            l[i + 1] += 1
        check(l)


def test_repeated_modified_jacobian_doubling(shim):
    """k modified-Jacobian doublings (curve.hpp mjac_dbl, W = a Z^4 carried along) == k jac_dbl == the oracle's
    2^k P, on the curve (a = 1) and on the c-model of an x-only key (a = c^2), from a random projective form,
    up to 64 doublings in a row (magnitudes must not grow: the shim runs with the limb bound checks on)."""
    rng = random.Random(41)
    for trial in range(6):
        pt = R.ec_mult(rng.randrange(1, N), tuple(R.EC_GEN))
        a, b = 1, R.BETA
        if trial % 2:  # the model y'^2 = x'^3 + c^2 x' + beta c^3 of the same point
            c = pt[1] * pt[1] % P
            pt = (c * pt[0] % P, c * c % P * 1 % P * 1)  # (c x, c^2 * (y / sqrt c)) with sqrt c = y
            a = c * c % P
        for k in (1, 4, 64 if trial < 2 else 7):
            outs = [(ctypes.c_uint32 * 8)() for _ in range(4)]
            shim.t_repeated_doubling(W(pt[0]), W(pt[1]), W(rng.randrange(1, P)), W(a), k, *outs)
            want = pt
            for _ in range(k):
                want = R.ec_double(want, a)
            assert (I(outs[0]), I(outs[1])) == want, (trial, k, "mjac")
            assert (I(outs[2]), I(outs[3])) == want, (trial, k, "jac")


def test_fe_inv_lehmer(shim):
    """The double-steered (Lehmer, nearest-integer quotients) inversion of the latency path: random values,
    the edge values that need the divsteps fallback (a partial quotient above 2^27: small x, p - small,
    powers of two), and one batch looked at from outside: unimodular integer rows below 2^29 that leave
    a last remainder at least 22 bits below the larger input."""
    rng = random.Random(11)
    out = (ctypes.c_uint32 * 8)()
    vals = [1, 2, 3, 4, 5, P - 1, P - 2, 2**251, 2**192 + 1, (P + 1) // 2, 2**29, 2**29 - 1, 2**58 + 1, P - 2**29]
    vals += [2**k for k in range(0, 252, 7)] + [P - 2**k for k in range(1, 251, 11)]
    vals += [rng.randrange(1, P) for _ in range(6000)] + [rng.randrange(1, 2**k) for k in (30, 53, 54, 64, 100, 128, 200) for _ in range(60)]
    vals += [P // k for k in range(2, 40)] + [P // k + 1 for k in range(2, 40)]
    for a in vals:
        shim.t_fe_inv_plain_lehmer(W(a), out)
        assert I(out) == pow(a, -1, P), hex(a)
    shim.t_fe_inv_plain_lehmer(W(0), out)
    assert I(out) == 0
    for a in vals[:60] + vals[-200:]:                  # fe_inv: Montgomery form, unreduced representatives
        for k in (0, 3, -2, 9):
            shim.t_fe_inv_lehmer_lazy(W(a), k, out)
            assert I(out) == pow(a, -1, P), (hex(a), k)
    for k in (0, 1, 5, -3):
        shim.t_fe_inv_lehmer_lazy(W(0), k, out)      # any representative of zero answers zero
        assert I(out) == 0, k
    rows = (ctypes.c_double * 4)()
    for _ in range(500):
        a = rng.randrange(2**200, 2**252)
        b = rng.randrange(2**199, a)
        assert shim.t_lehmer_batch(W(a), W(b), rows) == 1
        ua, va, ub, vb = (int(v) for v in rows)
        assert all(float(int(v)) == v for v in rows) and max(abs(ua), abs(va), abs(ub), abs(vb)) < 2**29
        assert abs(ua * vb - va * ub) == 1
        na, nb = ua * a + va * b, ub * a + vb * b
        assert abs(nb) < a >> 22 and abs(na) < a      # B is the last remainder: below (2^-27 + 2^-23) A on the true integers
    assert shim.t_lehmer_batch(W(P), W(1), rows) == 0          # partial quotient 2^251: not representable
    assert shim.t_lehmer_batch(W(12345), W(0), rows) == 1 and list(rows) == [1.0, 0.0, 0.0, 1.0]


def test_lehmer_with_an_imprecise_reciprocal(tmp_path):
    """the device steers Euclid with the bare v_rcp_f64, the host tests with an exact 1.0 / b.  Second
    build of the shim whose reciprocal carries a chosen relative error (alternating sign): every result stays
    exact (a wrong quotient costs progress only; SP_CHECK_BOUNDS watches the cofactors), the batch count stays
    inside the 24 of the loop for errors far beyond the instruction's (2^-24 ... 2^-12), and a value that does run
    out of batches is REPORTED (divsteps fallback), not answered from a stale remainder."""
    so = str(tmp_path / "host_shim_rcp.so")
    subprocess.check_call(["g++", "-O2", "-shared", "-fPIC", "-DSP_LEHMER_RCP_ERROR=1", "-o", so,
                           os.path.join(HERE, "host", "host_shim.cpp")])
    lib = ctypes.CDLL(so)
    lib.t_set_rcp_error.argtypes = [ctypes.c_double]
    rng = random.Random(12)
    out = (ctypes.c_uint32 * 8)()
    vals = [rng.randrange(1, P) for _ in range(1500)] + [P // k + 1 for k in range(2, 60)] + \
        [rng.randrange(1, 2**k) for k in (64, 128, 200, 251) for _ in range(50)] + [P - 1, P - 2, (P + 1) // 2]
    worst = {}
    for err in (0.0, 2.0**-40, 2.0**-24, 2.0**-20, 2.0**-12, 2.0**-3):
        lib.t_set_rcp_error(err)
        lib.t_take_lehmer_batches()
        most = 0
        for a in vals:
            lib.t_fe_inv_plain_lehmer(W(a), out)
            assert I(out) == pow(a, -1, P), (hex(a), err)
            most = max(most, lib.t_take_lehmer_batches())
        worst[err] = most
    assert worst[0.0] <= 13 and worst[2.0**-24] <= 14 and worst[2.0**-12] <= 24, worst
    print("lehmer batches at most, by reciprocal error:", worst)
    # running out of batches on purpose (a budget of 8 where a 252-bit inversion needs ~10): the value is REPORTED
    # as not converged - round 3 answered 0 from a stale remainder - and the public form takes the divsteps path
    lib.t_set_rcp_error(2.0**-24)
    lib.t_set_lehmer_budget(8)
    reported = 0
    for a in vals[:300]:
        reported += 1 - lib.t_lehmer_bezout_ok(W(a))
        lib.t_fe_inv_plain_lehmer(W(a), out)
        assert I(out) == pow(a, -1, P), hex(a)
    assert reported > 200, reported
    lib.t_set_lehmer_budget(24)
    assert all(lib.t_lehmer_bezout_ok(W(a)) == 1 for a in vals[:300])


def test_extreme_limb_patterns_keep_every_bound(shim):
    """Every pair of the extreme-limb-pattern felts (tests/workloads.py extreme_felts: all-ones limbs, p - small,
    2^k at the limb boundaries) through the bound-checked build of the multiplier, the lazy expression and the XYZZ
    chain: a limb or column that leaves its budget aborts the process, a wrong value fails the comparison.  The
    chord rule does not use the curve equation, so "points" with extreme coordinates exercise the addition formulas
    as they stand."""
    import sys
    sys.path.insert(0, HERE)
    import workloads as wl
    ext = wl.extreme_felts()
    out = (ctypes.c_uint32 * 8)()
    for a in ext:
        shim.t_fe_sqr(W(a), out)
        assert I(out) == a * a % P
        for b in ext:
            shim.t_fe_mul(W(a), W(b), out)
            assert I(out) == a * b % P, (hex(a), hex(b))
    rng = random.Random(19)
    for _ in range(4000):
        a, b, c, d, e, f = (rng.choice(ext) for _ in range(6))
        shim.t_fe_expr(W(a), W(b), W(c), W(d), W(e), W(f), out)
        assert I(out) == ((a - b) * (c + d) - e * f) % P
    x, y = (ctypes.c_uint32 * 8)(), (ctypes.c_uint32 * 8)()

    def chord(p, q):
        lam = (q[1] - p[1]) * pow(q[0] - p[0], -1, P) % P
        x3 = (lam * lam - p[0] - q[0]) % P
        return x3, (lam * (p[0] - x3) - p[1]) % P
    done = 0
    while done < 300:
        n = rng.choice((2, 3, 5, 9))
        pts = [(rng.choice(ext), rng.choice(ext)) for _ in range(n)]
        exp, ok = pts[0], True
        for q in pts[1:]:
            if (q[0] - exp[0]) % P == 0:
                ok = False
                break
            exp = chord(exp, q)
        if not ok:
            continue
        shim.t_xyzz_chain(WN([p[0] for p in pts]), WN([p[1] for p in pts]), n, x, y)
        assert (I(x), I(y)) == exp
        done += 1

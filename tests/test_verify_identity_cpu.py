"""The algebra behind csrc/ecdsa.hip verify_finish, checked with Python integers (no GPU, no library):

  * the acceptance test with the denominators cleared - En^2 == 4 P Q for x-only keys, En == -2 (Y_A ZZZ_A)
    (Y_B Z_B b) for point keys - holds exactly when r is x(A + B) (resp. x(A +- B)), for every projective
    representative of A (XYZZ) and B (Jacobian, on the c-model of an x-only key);
  * for a key whose c = x^3 + x + beta is a NON-residue the x-only identity has no solution r at all (4 P Q is a
    non-residue), which is why the ladder kernel carries no Legendre test of its own.

The group law comes from oracle/ref_py.py (math_utils.py:59-100 restated)."""
import random

from oracle import ref_py as R

P, N, BETA = R.FIELD_PRIME, R.EC_ORDER, R.BETA


def legendre(v):
    return pow(v % P, (P - 1) // 2, P)


def xyzz_of(pt, lam):
    x, y = pt
    zz, zzz = lam * lam % P, lam * lam * lam % P
    return x * zz % P, y * zzz % P, zz, zzz


def jac_of(pt, z):
    x, y = pt
    return x * z * z % P, y * z * z * z % P, z


def en_terms(A, B, c, r):
    """En, P, Q, Dn, a, b of verify_finish for A = (X, Y, ZZ, ZZZ), B = (X, Y, Z) on the c-model."""
    XA, YA, a, ZZZA = A
    XB, YB, ZB = B
    b = c * ZB * ZB % P
    xan, xbn = XA * b % P, XB * a % P
    Dn = (xan - xbn) % P
    s = (r * a % P * b + xan + xbn) % P
    Pp = YA * YA % P * pow(b, 3, P) % P
    Q = YB * YB % P * pow(ZZZA, 2, P) % P  # a^3 = ZZZ_A^2
    En = (s * Dn % P * Dn - Pp - Q) % P
    return En, Pp, Q, Dn, a, b


def test_projective_acceptance_identity_matches_the_group_law():
    rng = random.Random(17)
    G = tuple(R.EC_GEN)
    for it in range(12):
        A = R.ec_mult(rng.randrange(1, N), G)
        Bp = R.ec_mult(rng.randrange(1, N), G)  # the key's multiple on the curve itself
        lam, z = rng.randrange(1, P), rng.randrange(1, P)
        plus, minus = R.ec_add(A, Bp)[0], R.ec_add(A, R.ec_neg(Bp))[0]
        # point key: c = 1, B in Jacobian coordinates on the curve
        Aj, Bj = xyzz_of(A, lam), jac_of(Bp, z)
        for r, want_pt, want_x in ((plus, True, True), (minus, False, True), (rng.randrange(P), False, False)):
            En, Pp, Q, Dn, a, b = en_terms(Aj, Bj, 1, r)
            assert Dn != 0 and a * b % P != 0
            rhs = Aj[1] * Aj[3] % P * (Bj[1] * Bj[2] % P * b % P) % P
            assert (En == (-2 * rhs) % P) == want_pt
            assert (En * En % P == 4 * Pp * Q % P) == want_x
        # x-only key: the same point on the model y'^2 = x'^3 + c^2 x' + beta c^3 (x' = c x, y' = c^2 y / sqrt c)
        x, y = Bp
        c = (x * x * x + x + BETA) % P
        assert c == y * y % P
        # a multiple k * (x, sqrt c) of the key has model coordinates (c X, c^2 Y / sqrt c); with y = sqrt c chosen
        # as the key's own y the model point of Bp is (c x, c^2 * 1) scaled by the multiple - build it directly:
        k = rng.randrange(1, N)
        Bk = R.ec_mult(k, Bp)
        t = Bk[1] * pow(y, -1, P) % P                     # Y = t sqrt c with sqrt c = y
        model = (c * Bk[0] % P, c * c % P * t % P)
        assert (model[1] * model[1] - (pow(model[0], 3, P) + c * c % P * model[0] + BETA * pow(c, 3, P))) % P == 0
        Bm = jac_of(model, z)
        plus, minus = R.ec_add(A, Bk)[0], R.ec_add(A, R.ec_neg(Bk))[0]
        for r, want in ((plus, True), (minus, True), (rng.randrange(P), False)):
            En, Pp, Q, Dn, a, b = en_terms(Aj, Bm, c, r)
            assert (En * En % P == 4 * Pp * Q % P) == want


def test_non_residue_keys_can_never_be_accepted():
    """c a non-residue: the model is the quadratic twist; 4 P Q = (2 Y_A Y_B)^2 a^3 c^3 Z_B^6 is a non-residue for
    every point of it, so En^2 == 4 P Q has no solution r.  Also: the twist has odd order (no point with y = 0)."""
    rng = random.Random(23)
    assert (2 * P + 2 - N) % 2 == 1
    G = tuple(R.EC_GEN)
    done = 0
    while done < 8:
        x = rng.randrange(P)
        c = (x * x * x + x + BETA) % P
        if legendre(c) != P - 1:
            continue
        done += 1
        alpha_m, beta_m = c * c % P, BETA * pow(c, 3, P) % P
        base = (c * x % P, c * c % P)
        assert (base[1] ** 2 - (base[0] ** 3 + alpha_m * base[0] + beta_m)) % P == 0
        Bm = R.ec_mult(rng.randrange(2, 2**200), base, alpha_m)
        assert (Bm[1] ** 2 - (Bm[0] ** 3 + alpha_m * Bm[0] + beta_m)) % P == 0 and Bm[1] != 0
        A = R.ec_mult(rng.randrange(1, N), G)
        Aj, Bj = xyzz_of(A, rng.randrange(1, P)), jac_of(Bm, rng.randrange(1, P))
        for r in (rng.randrange(P), A[0], Bm[0] * pow(c, -1, P) % P):
            En, Pp, Q, Dn, a, b = en_terms(Aj, Bj, c, r)
            assert Dn != 0
            assert legendre(4 * Pp * Q) == P - 1          # a non-residue: nothing squares to it
            assert En * En % P != 4 * Pp * Q % P

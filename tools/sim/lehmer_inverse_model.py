"""Model of the f64-guided (Lehmer, nearest-integer quotients) modular inversion that replaces the
divsteps batches on the latency path (csrc/fp29.hpp lehmer_*; csrc/quad.hpp fe_inv_plain_quad).

Big numbers are nine signed 29-bit limbs as on the device; every batch
  * turns A and B into doubles by Horner (53-bit relative approximations, no search for the top limb),
  * runs Euclid with nearest-integer quotients on the doubles while the divisor stays >= 2^-27 of the
    batch's larger input: the 2x2 cofactor matrix stays below 2^29 in magnitude (a batch that
    starts with B below 2^-27 A - a partial quotient no int32 matrix can hold, probability ~2^-27 per
    batch on random input - sends the value to the divsteps inversion instead),
  * applies the integer matrix to (A, B) exactly and to (D, E) modulo p with the multiple of p
    estimated in doubles.
Any quotient sequence gives a unimodular matrix, so the doubles only steer; the model checks the
inverse, counts batches / Euclid steps and records the largest matrix entry and p-multiple.
Run: python tools/sim/lehmer_inverse_model.py [count]"""
import random
import sys

P = 2**251 + 17 * 2**192 + 1
LB = 29
MASK = (1 << LB) - 1
QMAX = 2.0**27


def to_limbs(v):
    """signed-top limb form: low 8 limbs in [0, 2^29), top limb signed"""
    l = []
    for _ in range(8):
        l.append(v & MASK)
        v >>= LB
    l.append(v)
    return l


def horner(l):
    s = float(l[8])
    for i in range(7, -1, -1):
        s = s * 536870912.0 + float(l[i])  # one fma on the device; here mul then add: model both roundings
    return s


def exact_fnma(q, b, a, clamped):
    r = int(a) - int(q) * int(b)
    rf = float(r)  # the device's fma: one rounding
    assert clamped or int(rf) == r, "remainder not exact"
    return rf


def euclid_batch(a, b, stats):
    """a, b >= 0 doubles.  Returns the rows (ua, va), (ub, vb) with new_a = ua*a + va*b, new_b = ub*a + vb*b
    (as integers on the true values), or None when the batch is not representable.  Remainders are signed
    (nearest-integer quotients): no absolute values inside the loop, exactly as lehmer_batch does it."""
    ua, va, ub, vb = 1.0, 0.0, 0.0, 1.0
    a0 = max(a, b)
    thresh = max(a0 * 2.0**-27, 0.5)
    if min(a, b) < a0 * 2.0**-27:
        return None  # a partial quotient above 2^27: the caller falls back to the divsteps inversion
    steps = 0
    while min(abs(a), abs(b)) >= thresh:
        q = float(round(a * (1.0 / b)))  # rndne of a * rcp(b)
        a = exact_fnma(q, b, a, False)
        ua, va = ua - q * ub, va - q * vb
        steps += 1
        if abs(a) >= thresh:             # the second step of a trip is a no-op otherwise
            q = float(round(b * (1.0 / a)))
            b = exact_fnma(q, a, b, False)
            ub, vb = ub - q * ua, vb - q * va
            steps += 1
        assert max(abs(ua), abs(va), abs(ub), abs(vb)) < 2**53
    stats["steps"] += steps
    stats["maxsteps"] = max(stats["maxsteps"], steps)
    if abs(a) < abs(b):                  # the smaller remainder is the new B
        ua, va, ub, vb = ub, vb, ua, va
    return ua, va, ub, vb


def inverse(x, stats):
    A, B, D, E = P, x, 0, 1
    batches = 0
    while True:
        la, lb = to_limbs(A), to_limbs(B)
        ad, bd = horner(la), horner(lb)
        if bd == 0.0:
            assert B == 0
            break
        sa, sb = (-1 if ad < 0 else 1), (-1 if bd < 0 else 1)
        rows = euclid_batch(abs(ad), abs(bd), stats)
        if rows is None:
            stats["fallbacks"] += 1
            return pow(x, -1, P)
        ua, va, ub, vb = rows
        ua, ub = int(ua) * sa, int(ub) * sa
        va, vb = int(va) * sb, int(vb) * sb
        m = max(abs(ua), abs(va), abs(ub), abs(vb))
        stats["maxcof"] = max(stats["maxcof"], m)
        assert m < 2**31
        dd, ed = horner(to_limbs(D)), horner(to_limbs(E))
        A, B = ua * A + va * B, ub * A + vb * B
        D2 = ua * D + va * E
        E2 = ub * D + vb * E
        td = round((float(ua) * dd + float(va) * ed) * (1.0 / float(P)))
        te = round((float(ub) * dd + float(vb) * ed) * (1.0 / float(P)))
        stats["maxt"] = max(stats["maxt"], abs(td), abs(te))
        D, E = D2 - td * P, E2 - te * P
        assert abs(D) < P and abs(E) < P, "cofactor reduction failed"
        assert abs(A) < 2**257 and abs(B) < 2**257
        batches += 1
        assert batches < 26
    stats["batches"] += batches
    stats["maxbatches"] = max(stats["maxbatches"], batches)
    if x % P == 0:
        return 0  # contract of fe_inv: 0 -> 0 (A = p, D = 0 at the first test)
    assert abs(A) == 1, "gcd is not one"
    return (D * A) % P


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
    rnd = random.Random(7)
    stats = dict(steps=0, maxsteps=0, batches=0, maxbatches=0, maxcof=0, maxt=0, fallbacks=0)
    xs = [1, 2, 3, P - 1, P - 2, (P + 1) // 2, 2**250, 2**251, 2**192, 17 * 2**192, P // 3]
    xs += [rnd.randrange(1, P) for _ in range(n)]
    xs += [rnd.randrange(1, 2**k) for k in (8, 29, 30, 53, 54, 58, 64, 128, 200) for _ in range(20)]
    xs += [P - rnd.randrange(1, 2**k) for k in (8, 29, 53, 64, 128) for _ in range(20)]
    for x in xs:
        inv = inverse(x, stats)
        assert inv * x % P == 1
    assert inverse(0, dict(stats)) == 0
    k = len(xs)
    print("values %d: batches mean %.2f max %d; euclid steps mean %.1f per inversion, max %d per batch; "
          "largest matrix entry 2^%.1f, largest p-multiple 2^%.1f, fallbacks %d"
          % (k, stats["batches"] / k, stats["maxbatches"], stats["steps"] / k, stats["maxsteps"],
             __import__("math").log2(stats["maxcof"]), __import__("math").log2(max(1, stats["maxt"])), stats["fallbacks"]))


if __name__ == "__main__":
    main()

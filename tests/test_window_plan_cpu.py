"""Integer / group model of the signed window plan of the Pedersen tables (csrc/context.hip), checked
against the oracle's pedersen_hash with the oracle's own affine arithmetic.  Pure Python, no GPU:
pins the identity the device tables rely on, for several window widths."""
import random

import pytest

from oracle import ref_py as R

P, N = R.FIELD_PRIME, R.EC_ORDER
BITS = 504  # x (252 bits) || y (252 bits)


def make_plan(log2e):
    """(start, bits) per window: as many (log2e + 1)-bit signed windows as fit, the remaining low
    bits (at least one) as the unsigned window 0 - context.hip make_plan."""
    sw = log2e + 1
    k = BITS // sw
    w0 = BITS - k * sw
    if w0 == 0:
        k, w0 = k - 1, sw
    return [(0, w0)] + [(w0 + g * sw, sw) for g in range(k)]


def test_plan_shapes():
    assert len(make_plan(21)) == 23 and make_plan(21)[0] == (0, 20)
    assert len(make_plan(26)) == 19 and make_plan(26)[0] == (0, 18)
    assert len(make_plan(27)) == 18 and make_plan(27)[0] == (0, 28)
    for log2e in range(4, 28):
        plan = make_plan(log2e)
        assert plan[0][1] >= 1 and sum(b for _, b in plan) == BITS
        assert all(plan[i][0] + plan[i][1] == plan[i + 1][0] for i in range(len(plan) - 1))


def per_bit_points():
    """C_j of the string x || y: the reference's CONSTANT_POINTS[2 + j]."""
    return [tuple(p) for p in R.CONSTANT_POINTS[2 : 2 + BITS]]


def half_points(c):
    """C'_j = C_j / 2: the previous point inside a doubling chain, (N + 1)/2 times the chain head."""
    half = (N + 1) // 2
    out = []
    for first, count in ((0, 248), (248, 4), (252, 248), (500, 4)):
        for j in range(count):
            out.append(R.ec_mult(half, c[first]) if j == 0 else c[first + j - 1])
    return out


def neg(pt):
    return (pt[0], (-pt[1]) % P)


@pytest.mark.parametrize("log2e", [4, 9, 21, 26, 27])
def test_sum_of_selected_entries_is_the_hash(log2e):
    c = per_bit_points()
    h = half_points(c)
    for a, b in zip(c[:3] + c[248:250], h[:3] + h[248:250]):
        assert R.ec_double(b) == a
    plan = make_plan(log2e)
    shift = tuple(R.SHIFT_POINT)
    # window offsets: O_0 = SHIFT + sum_{j >= w0} C'_j ;  O_g = C'_top - sum_{b < log2e} C'_{s+b}
    w0 = plan[0][1]
    o0 = shift
    for j in range(w0, BITS):
        o0 = R.ec_add(o0, h[j])
    offsets = [o0]
    for s0, _ in plan[1:]:
        o = h[s0 + log2e]
        for b in range(log2e):
            o = R.ec_add(o, neg(h[s0 + b]))
        offsets.append(o)
    rng = random.Random(log2e)
    cases = [(0, 0), (P - 1, P - 1), (1, 0), (0, 1), (2**248 - 1, 2**251), (2**251 + 5, 2**248)]
    cases += [(rng.randrange(P), rng.randrange(P)) for _ in range(4)]
    for x, y in cases:
        string = x | (y << 252)
        acc = None
        for g, (s0, nb) in enumerate(plan):
            raw = (string >> s0) & ((1 << nb) - 1)
            if g == 0:
                idx, negative = raw, False
            else:
                negative = (raw >> log2e) == 0
                idx = (~raw if negative else raw) & ((1 << log2e) - 1)
            entry = offsets[g]                       # T_g[idx] = O_g + sum_{b in idx} C_{s+b}
            for b in range(nb if g == 0 else log2e):
                if (idx >> b) & 1:
                    entry = R.ec_add(entry, c[s0 + b])
            if negative:
                entry = neg(entry)
            acc = entry if acc is None else R.ec_add(acc, entry)
        assert acc[0] == R.pedersen_hash(x, y), (log2e, hex(x), hex(y))

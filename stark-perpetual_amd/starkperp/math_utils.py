"""Python-int helpers with the names and contracts of the reference's
starkware/crypto/signature/math_utils.py (ECPoint, div_mod, ec_add, ec_neg, ec_double, ec_mult,
is_quad_residue, sqrt_mod).  They exist so that code importing them keeps working and to expand
the constant-point table at import; the hot functions of `signature` (pedersen_hash, sign, verify,
key derivation) do not use them - those run in libstarkperp on the GPU."""
from typing import Tuple

ECPoint = Tuple[int, int]


def div_mod(n: int, m: int, p: int) -> int:
    """0 <= x < p with (m * x) % p == n   (math_utils.py:50-56)."""
    try:
        inv = pow(m, -1, p)
    except ValueError:
        raise AssertionError()  # the reference asserts gcd == 1
    return (n * inv) % p


def ec_add(point1: ECPoint, point2: ECPoint, p: int) -> ECPoint:
    """math_utils.py:59-68: affine chord, x coordinates must differ."""
    assert (point1[0] - point2[0]) % p != 0
    m = div_mod(point1[1] - point2[1], point1[0] - point2[0], p)
    x = (m * m - point1[0] - point2[0]) % p
    y = (m * (point1[0] - x) - point1[1]) % p
    return x, y


def ec_neg(point: ECPoint, p: int) -> ECPoint:
    """math_utils.py:71-76."""
    x, y = point
    return (x, (-y) % p)


def ec_double(point: ECPoint, alpha: int, p: int) -> ECPoint:
    """math_utils.py:79-88: affine tangent, y != 0."""
    assert point[1] % p != 0
    m = div_mod(3 * point[0] * point[0] + alpha, 2 * point[1], p)
    x = (m * m - 2 * point[0]) % p
    y = (m * (point[0] - x) - point[1]) % p
    return x, y


def ec_mult(m: int, point: ECPoint, alpha: int, p: int) -> ECPoint:
    """math_utils.py:91-100 (0 < m < order).  Iterative form of the reference's recursion."""
    assert m >= 1
    deferred = []
    while m != 1:
        if m & 1:
            deferred.append(point)
            m -= 1
        else:
            point = ec_double(point, alpha, p)
            m >>= 1
    for q in reversed(deferred):
        point = ec_add(point, q, p)
    return point


def is_quad_residue(n: int, p: int) -> bool:
    """math_utils.py:36-40 (sympy semantics: 0 counts as a residue)."""
    n %= p
    return n == 0 or pow(n, (p - 1) // 2, p) == 1


def sqrt_mod(n: int, p: int) -> int:
    """Smallest m >= 0 with m*m % p == n (math_utils.py:43-47); Tonelli-Shanks."""
    n %= p
    if n == 0:
        return 0
    if not is_quad_residue(n, p):
        raise ValueError("not a quadratic residue")
    odd, twos = p - 1, 0
    while odd % 2 == 0:
        odd //= 2
        twos += 1
    g = 2
    while is_quad_residue(g, p):
        g += 1
    c = pow(g, odd, p)
    root = pow(n, (odd + 1) // 2, p)
    t = pow(n, odd, p)
    m = twos
    while t != 1:
        i, probe = 0, t
        while probe != 1:
            probe = probe * probe % p
            i += 1
        b = pow(c, 1 << (m - i - 1), p)
        root = root * b % p
        c = b * b % p
        t = t * c % p
        m = i
    return min(root, p - root)


def pi_as_string(digits: int) -> str:
    """math_utils.py:28-33: pi as decimal digits without the point ("314...").  The reference prints
    mpmath's pi at `digits` significant digits; here: Machin's formula in integers with guard
    digits, rounded to nearest at `digits` significant digits, trailing zeros dropped as mpmath's
    str() does.  mpmath rounds through a binary intermediate, so its LAST digit or two can differ
    from the correctly rounded decimal; the one caller (nothing_up_my_sleeve_gen.py:57) asks for
    100 spare digits for exactly that reason and reads only the leading ones."""
    assert digits >= 2
    guard = 12
    scale = 10 ** (digits - 1 + guard)

    def arctan_inverse(q):
        total = term = scale // q
        q2, k, sign = q * q, 3, -1
        while term:
            term //= q2
            total += sign * (term // k)
            k, sign = k + 2, -sign
        return total

    pi_scaled = 16 * arctan_inverse(5) - 4 * arctan_inverse(239)
    rounded = (pi_scaled + 10**guard // 2) // 10**guard  # `digits` significant digits
    text = str(rounded)
    fraction = text[1:].rstrip("0") or "0"
    return "3" + fraction

"""Integer model of the signed comb used by the ECDSA key tables (csrc/ecdsa.hip "Key tables" /
comb_mul / key_table_kernel): the recoding identity and the absence of self-meeting additions.
Pure Python, no GPU."""
import random

from oracle import ref_py as R

N = R.EC_ORDER


def table_multiple(v):
    """Integer multiple of Q held by table entry v: row 7 positive, row i < 7 signed by bit i."""
    return 2**224 + sum((1 if (v >> i) & 1 else -1) * 2 ** (32 * i) for i in range(7))


def comb_columns(k):
    """(sign, index) per column 31..0 for an odd k, exactly as comb_mul reads them off
    E = (k - 1)/2 + 2^255 (word i of E = comb row i, bit c = column c)."""
    assert k % 2 == 1
    e = (k - 1) // 2 + 2**255
    out = []
    for col in range(31, -1, -1):
        bits = [(e >> (32 * i + col)) & 1 for i in range(8)]
        idx = sum(bits[i] << i for i in range(7))
        negative = bits[7] == 0
        if negative:
            idx ^= 127
        out.append((-1 if negative else 1, idx))
    return out


def test_comb_recoding_reconstructs_the_scalar():
    rng = random.Random(8)
    for k in [1, 3, N - 2, N - 4, 2**251 + 1] + [rng.randrange(N) | 1 for _ in range(200)]:
        if k >= N:
            continue
        acc = 0
        cols = comb_columns(k)
        assert cols[0][0] == 1  # bit 255 of E is always set: the top column is +T
        for sign, idx in cols:
            acc = 2 * acc + sign * table_multiple(idx)
        assert acc == k


def test_gray_walk_visits_every_entry_once():
    """key_table_kernel: thread t starts at entry t << 5 and flips row ctz(g) at step g."""
    seen = set()
    for t in range(4):
        gray, value = 0, table_multiple(t << 5)
        seen.add((t << 5) | gray)
        for g in range(1, 32):
            bit = (g & -g).bit_length() - 1
            gray ^= 1 << bit
            value += (2 if (gray >> bit) & 1 else -2) * 2 ** (32 * bit)
            assert value == table_multiple((t << 5) | gray)
            seen.add((t << 5) | gray)
    assert seen == set(range(128))


def test_no_scalar_meets_its_own_table_entry():
    """Why the comb needs no exceptional-case branch for honest tables (the kernel keeps a guard
    anyway): the last column adds a = +-t_v to 2m with 2m + a = k, and 2m == +-a (mod N) would need
    k == 2a or k == 0 (mod N).  For each of the 256 signed table multiples, k = 2a mod N is either
    even (recoded as N - k) or its own column-0 digit is a different multiple."""
    hits = 0
    for v in range(128):
        t = table_multiple(v)
        for a in (t, -t):
            k = 2 * a % N
            if k % 2 == 0:
                continue  # the kernel works on N - k; covered by the other sign
            sign, idx = comb_columns(k)[-1]  # column 0, the last addition
            if sign * table_multiple(idx) == a:
                hits += 1
    assert hits == 0



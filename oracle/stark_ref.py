"""
ORACLE (test infrastructure, NOT product code) for the prover-side half of the hot path:
radix-2 NTT / coset LDE, the build-defined Pedersen-step AIR, FRI folding and Pedersen-Merkle
commitments over p = 2^251 + 17*2^192 + 1.

PARITY UNPINNED: the reference tree contains no prover (no NTT, AIR or FRI code - SURVEY.md
section 2, rows A10-A13), so there is nothing to pin these functions against.  What IS tied to the
reference: the field and its generator (pedersen_params.json:20-21), the step relation of the AIR
(derived line by line from signature.py:305-317 and math_utils.py:64-67) and every hash inside a
commitment (oracle.ref_py.pedersen_hash, pinned).  Correctness of the rest is established by
algebraic self-checks in tests/test_stark_oracle.py (naive DFT, iNTT o NTT = id, fold vs direct
polynomial evaluation, the composition polynomial of a valid trace has degree < 4n - deg Z_H).

Plain Python big-int arithmetic, written for clarity at small sizes.
"""
from . import ref_py as R

P = R.FIELD_PRIME
GEN = R.FIELD_GEN  # 3 generates GF(p)^*;  p - 1 = 2^192 * (2^59 + 17)
BLOWUP = 4
ROWS_PER_HASH = 512  # two 256-row blocks (252 bit steps + 4 padding rows) per hash


def root_of_unity(log_n):
    """Primitive 2^log_n-th root of unity: 3^((p-1)/2^log_n)."""
    assert 0 <= log_n <= 192
    return pow(GEN, (P - 1) >> log_n, P)


def naive_dft(coeffs, w):
    n = len(coeffs)
    return [sum(c * pow(w, i * k, P) for k, c in enumerate(coeffs)) % P for i in range(n)]


def ntt(coeffs, w):
    """Evaluations of sum c_k x^k at x = w^i, i = 0..n-1 (natural order both sides)."""
    n = len(coeffs)
    if n == 1:
        return list(coeffs)
    even = ntt(coeffs[0::2], w * w % P)
    odd = ntt(coeffs[1::2], w * w % P)
    out = [0] * n
    t = 1
    for i in range(n // 2):
        o = t * odd[i] % P
        out[i] = (even[i] + o) % P
        out[i + n // 2] = (even[i] - o) % P
        t = t * w % P
    return out


def intt(evals, w):
    n = len(evals)
    inv_n = pow(n, -1, P)
    return [v * inv_n % P for v in ntt(evals, pow(w, -1, P))]


def lde(evals, blowup=BLOWUP, shift=GEN):
    """Values of the interpolant of `evals` (over <w_n>) on the coset shift * <w_{blowup n}>."""
    n = len(evals)
    log_n = n.bit_length() - 1
    coeffs = intt(evals, root_of_unity(log_n))
    shifted = [c * pow(shift, k, P) % P for k, c in enumerate(coeffs)] + [0] * (n * (blowup - 1))
    return ntt(shifted, root_of_unity(log_n + blowup.bit_length() - 1))


# ---- the Pedersen-step AIR ----------------------------------------------------------------------
# Columns s, px, py, lam.  A hash occupies 512 rows: block 0 (rows 0..255) consumes x, block 1
# (rows 256..511) consumes y.  In row j of a block (j = 0..251) with constant point
# C = CONSTANT_POINTS[2 + 252*block + j]:   b = s - 2 s_next  in {0,1};  if b: (px,py)_next =
# (px,py) + C by the chord rule with slope lam;  else unchanged.  Rows 252..255 are padding (s = 0).
def pedersen_trace(inputs):
    """inputs: list of (x, y).  Returns columns [s, px, py, lam], each of len 512 * len(inputs)."""
    s_col, px_col, py_col, lam_col = [], [], [], []
    for x, y in inputs:
        acc = tuple(R.SHIFT_POINT)
        for block, elem in enumerate((x, y)):
            s = elem
            for j in range(256):
                s_col.append(s)
                px_col.append(acc[0])
                py_col.append(acc[1])
                lam = 0
                if j < 252:
                    c = R.CONSTANT_POINTS[2 + 252 * block + j]
                    if s & 1:
                        lam = R.div_mod(acc[1] - c[1], acc[0] - c[0], P)
                        acc = R.ec_add(acc, c)
                    s >>= 1
                lam_col.append(lam)
    return [s_col, px_col, py_col, lam_col]


def periodic_columns():
    """Period-512 columns: cx, cy (constant points; padding rows repeat the generator), and the
    selectors  step (1 except on the last row of each 256-block), mid (row 255), end (row 511),
    zero252 (rows 252 and 508: s must be 0 there)."""
    cx, cy, step, mid, end, z252 = [], [], [], [], [], []
    for r in range(512):
        block, j = divmod(r, 256)
        c = R.CONSTANT_POINTS[2 + 252 * block + j] if j < 252 else R.EC_GEN
        cx.append(c[0])
        cy.append(c[1])
        step.append(0 if j == 255 else 1)
        mid.append(1 if r == 255 else 0)
        end.append(1 if r == 511 else 0)
        z252.append(1 if j == 252 else 0)
    return [cx, cy, step, mid, end, z252]


N_CONSTRAINTS = 11


def constraint_values(cur, nxt, per):
    """The eleven constraint polynomials at one point.  cur / nxt = (s, px, py, lam) at x and at
    w_n x;  per = (cx, cy, step, mid, end, z252) at x."""
    s, px, py, lam = cur
    s_n, px_n, py_n, _ = nxt
    cx, cy, step, mid, end, z252 = per
    b = (s - 2 * s_n) % P
    nb = (1 - b) % P
    sx, sy = R.SHIFT_POINT
    return [
        step * b * (b - 1) % P,
        step * b * (lam * (px - cx) - (py - cy)) % P,
        step * b * (lam * lam - px - cx - px_n) % P,
        step * b * (lam * (px - px_n) - py - py_n) % P,
        step * nb * (px_n - px) % P,
        step * nb * (py_n - py) % P,
        mid * (px_n - px) % P,
        mid * (py_n - py) % P,
        end * (px_n - sx) % P,
        end * (py_n - sy) % P,
        z252 * s % P,
    ]


# ---- the EC-ladder AIR (the ECDSA builtin's building block) --------------------------------------
# One instance = one mimic_ec_mult_air(m, point, shift_point) call (signature.py:176-190), 256 rows.
# Columns m, px, py, qx, qy, la, ld.  Row j = 0..250:  b = m - 2 m_next in {0,1};  the doubling
# point (qx, qy) is doubled on EVERY row by the tangent rule with slope ld (math_utils.py:79-88);
# if b the partial sum (px, py) takes the chord step with slope la (math_utils.py:59-68), else it
# is held.  Row 0 starts from the shift point, m = 0 at row 251, rows 251..254 keep doubling with
# b = 0, row 255 is free (block boundary).  (px, py) at row 251 is m * point + shift_point.
def ec_ladder_trace(inputs, shift_point=None):
    """inputs: list of (m, (qx, qy)) with 0 < m < 2^251.  Returns 7 columns of 256 rows each."""
    shift_point = tuple(shift_point or R.SHIFT_POINT)
    cols = [[] for _ in range(7)]
    for m, point in inputs:
        acc, q = shift_point, tuple(point)
        for j in range(256):
            la = ld = 0
            bit = m & 1 if j < 251 else 0
            row = [m, acc[0], acc[1], q[0], q[1]]
            if j < 255:
                ld = R.div_mod(3 * q[0] * q[0] + R.ALPHA, 2 * q[1], P)
                if bit:
                    la = R.div_mod(acc[1] - q[1], acc[0] - q[0], P)
                    acc = R.ec_add(acc, q)
                q = R.ec_double(q)
                m >>= 1
            for c, v in zip(cols, row + [la, ld]):
                c.append(v)
    return cols


def ec_ladder_periodic_columns(shift_point=None):
    """Period-256 selectors: step (rows 0..254), first (row 0), z251 (row 251)."""
    step = [0 if j == 255 else 1 for j in range(256)]
    first = [1 if j == 0 else 0 for j in range(256)]
    z251 = [1 if j == 251 else 0 for j in range(256)]
    return [step, first, z251]


N_EC_LADDER_CONSTRAINTS = 12


def ec_ladder_constraint_values(cur, nxt, per, shift_point=None):
    sx, sy = tuple(shift_point or R.SHIFT_POINT)
    m, px, py, qx, qy, la, ld = cur
    m_n, px_n, py_n, qx_n, qy_n, _, _ = nxt
    step, first, z251 = per
    b = (m - 2 * m_n) % P
    nb = (1 - b) % P
    return [
        step * b * (b - 1) % P,
        step * (ld * 2 * qy - 3 * qx * qx - R.ALPHA) % P,
        step * (qx_n - ld * ld + 2 * qx) % P,
        step * (qy_n - ld * (qx - qx_n) + qy) % P,
        step * b * (la * (px - qx) - (py - qy)) % P,
        step * b * (px_n - la * la + px + qx) % P,
        step * b * (py_n - la * (px - px_n) + py) % P,
        step * nb * (px_n - px) % P,
        step * nb * (py_n - py) % P,
        first * (px - sx) % P,
        first * (py - sy) % P,
        z251 * m % P,
    ]


# ---- the ECDSA-verification AIR: what verify() mimics, signature.py:217-260 -----------
# One verification = 1024 rows of the ten columns m, px, py, qx, qy, la, ld, cx, cy, cr:
#   rows   0..255   ladder z * G   from MINUS_SHIFT_POINT (signature.py:252)
#   rows 256..511   ladder r * Q   from SHIFT_POINT       (:253)
#   rows 512..767   ladder w * B   from SHIFT_POINT, B = zG + rQ (:254: the base point of this ladder is
#                   tied to the two outputs by the chord rule at row 511)
#   row  767        x(wB - SHIFT_POINT) == r              (:255)
#   rows 768..1023  idle (keeps the period a power of two)
# Every ladder is the EC-ladder AIR above (doubling on every row, chord step when the bit is set, m = 0
# at row 251 - which is also the 251-bit range check of z, r and w, :225-227).  cx, cy carry the output of
# ladder 0 through block 1; cr carries r from row 256 to row 767.  Q must be on the curve (:241).
# w = s^-1 mod N (:220) is an INPUT of the AIR exactly as in the reference, whose `verify` computes it
# before "mimicking the AIR"; ecdsa_instance() checks it natively.
def ecdsa_instance(z, r, s, pubkey):
    """(z, r, w, Q) of a signature the reference accepts; raises AssertionError otherwise."""
    w = R.inv_mod_curve_size(s)
    assert 1 <= r < 2**251 and 1 <= w < 2**251 and 0 < z < 2**251 and R.is_point_on_curve(*pubkey)
    assert w * s % R.EC_ORDER == 1
    assert R.verify(z, r, s, tuple(pubkey))
    return z, r, w, tuple(pubkey)


def ecdsa_trace(instances):
    """instances: list of (z, r, w, (Qx, Qy)).  Returns 10 columns of 1024 rows each."""
    cols = [[] for _ in range(10)]
    shift, mshift = tuple(R.SHIFT_POINT), tuple(R.MINUS_SHIFT_POINT)
    for z, r, w, q in instances:
        lad0 = ec_ladder_trace([(z, tuple(R.EC_GEN))], mshift)
        lad1 = ec_ladder_trace([(r, q)], shift)
        zg = (lad0[1][251], lad0[2][251])
        rq = (lad1[1][251], lad1[2][251])
        b = R.ec_add(zg, rq)
        lad2 = ec_ladder_trace([(w, b)], shift)
        wb = (lad2[1][251], lad2[2][251])
        lad1[5][255] = R.div_mod(rq[1] - zg[1], rq[0] - zg[0], P)            # slope of zG + rQ at row 511
        lad2[5][255] = R.div_mod(wb[1] + shift[1], wb[0] - shift[0], P)      # slope of wB + (-shift) at row 767
        assert (lad2[5][255] ** 2 - wb[0] - shift[0]) % P == r, "not a valid signature"
        for block, lad in enumerate((lad0, lad1, lad2)):
            for c in range(7):
                cols[c] += lad[c]
            cols[7] += [zg[0] if block == 1 else 0] * 256
            cols[8] += [zg[1] if block == 1 else 0] * 256
            cols[9] += [r if block in (1, 2) else 0] * 256
        for c in range(10):
            cols[c] += [0] * 256
    return cols


def ecdsa_periodic_columns():
    """Period-1024 tables: step, first, start_y, z251, gbase, oncurve, carry_load, carry_hold, addb, rload,
    rhold, fin."""
    def rows(fn):
        return [1 if fn(i) else 0 for i in range(1024)]
    ladder = lambda i: i < 768
    step = rows(lambda i: ladder(i) and i % 256 != 255)
    first = rows(lambda i: ladder(i) and i % 256 == 0)
    start_y = [0] * 1024
    start_y[0], start_y[256], start_y[512] = R.MINUS_SHIFT_POINT[1], R.SHIFT_POINT[1], R.SHIFT_POINT[1]
    return [step, first, start_y, rows(lambda i: ladder(i) and i % 256 == 251), rows(lambda i: i == 0),
            rows(lambda i: i == 256), rows(lambda i: i == 255), rows(lambda i: 256 <= i < 511),
            rows(lambda i: i == 511), rows(lambda i: i == 256), rows(lambda i: 256 <= i < 767), rows(lambda i: i == 767)]


N_ECDSA_CONSTRAINTS = 26


def ecdsa_constraint_values(cur, nxt, per):
    sx, sy = R.SHIFT_POINT
    gx, gy = R.EC_GEN
    m, px, py, qx, qy, la, ld, cx, cy, cr = cur
    m_n, px_n, py_n, qx_n, qy_n, _, _, cx_n, cy_n, cr_n = nxt
    step, first, start_y, z251, gbase, oncurve, cload, chold, addb, rload, rhold, fin = per
    b = (m - 2 * m_n) % P
    nb = (1 - b) % P
    return [
        step * b * (b - 1) % P,
        step * (ld * 2 * qy - 3 * qx * qx - R.ALPHA) % P,
        step * (qx_n - ld * ld + 2 * qx) % P,
        step * (qy_n - ld * (qx - qx_n) + qy) % P,
        step * b * (la * (px - qx) - (py - qy)) % P,
        step * b * (px_n - la * la + px + qx) % P,
        step * b * (py_n - la * (px - px_n) + py) % P,
        step * nb * (px_n - px) % P,
        step * nb * (py_n - py) % P,
        first * (px - sx) % P,
        (first * py - start_y) % P,
        z251 * m % P,
        gbase * (qx - gx) % P,
        gbase * (qy - gy) % P,
        oncurve * (qy * qy - qx * qx * qx - R.ALPHA * qx - R.BETA) % P,
        cload * (cx_n - px) % P,
        cload * (cy_n - py) % P,
        chold * (cx_n - cx) % P,
        chold * (cy_n - cy) % P,
        addb * (la * (px - cx) - (py - cy)) % P,
        addb * (qx_n - la * la + px + cx) % P,
        addb * (qy_n - la * (px - qx_n) + py) % P,
        rload * (cr - m) % P,
        rhold * (cr_n - cr) % P,
        fin * (la * (px - sx) - (py + sy)) % P,
        fin * (cr - la * la + px + sx) % P,
    ]


# ---- range-check AIR ------------------------------------------------------------------------------
# 0 <= value < 2^128 by bit decomposition (the statement of the Cairo range-check builtin, which bounds the
# amounts / ids / nonces / timestamps of the exchange messages: perpetual_messages.py:226-236 asserts the same
# bounds on the Python side).  One column, 128 rows per value: v_i = value >> i.
RANGE_CHECK_BITS = 128


def range_check_trace(values):
    col = []
    for v in values:
        col.extend(v >> i for i in range(RANGE_CHECK_BITS))
    return [col]


def range_check_periodic_columns():
    return [[1] * 127 + [0], [0] * 127 + [1]]


N_RANGE_CHECK_CONSTRAINTS = 2


def range_check_constraint_values(cur, nxt, per):
    v, vn = cur[0], nxt[0]
    step, last = per
    b = (v - 2 * vn) % P
    return [step * b * (b - 1) % P, last * v * (v - 1) % P]


AIRS = {
    "ecdsa": {"n_cols": 10, "period": 1024, "n_constraints": N_ECDSA_CONSTRAINTS,
              "periodic": ecdsa_periodic_columns, "constraints": ecdsa_constraint_values},
    "pedersen": {"n_cols": 4, "period": 512, "n_constraints": N_CONSTRAINTS,
                 "periodic": periodic_columns, "constraints": constraint_values},
    "ec_ladder": {"n_cols": 7, "period": 256, "n_constraints": N_EC_LADDER_CONSTRAINTS,
                  "periodic": ec_ladder_periodic_columns, "constraints": ec_ladder_constraint_values},
    "range_check": {"n_cols": 1, "period": 128, "n_constraints": N_RANGE_CHECK_CONSTRAINTS,
                    "periodic": range_check_periodic_columns, "constraints": range_check_constraint_values},
}


def composition_on_coset(trace_lde, per_lde, n, alphas, shift=GEN, air="pedersen"):
    """Random linear combination of the constraints divided by Z_H(x) = x^n - 1, on the LDE coset.
    trace_lde: columns of 4n values;  per_lde: tables of 4*period values (index i mod 4*period)."""
    spec = AIRS[air]
    m = BLOWUP * n
    w = root_of_unity(m.bit_length() - 1)
    out = []
    zinv = [pow((pow(shift, n, P) * pow(w, n * k, P) - 1) % P, -1, P) for k in range(BLOWUP)]
    for i in range(m):
        cur = [col[i] for col in trace_lde]
        nxt = [col[(i + BLOWUP) % m] for col in trace_lde]
        per = [t[i % (BLOWUP * spec["period"])] for t in per_lde]
        cv = spec["constraints"](cur, nxt, per)
        acc = sum(a * c for a, c in zip(alphas, cv)) % P
        out.append(acc * zinv[i % BLOWUP] % P)
    return out


def periodic_lde(n, shift=GEN, air="pedersen"):
    """Periodic columns evaluated on the LDE coset: q(x^(n/period)) with x = shift * w_{4n}^i takes
    4*period distinct values = the blowup-4 LDE of the column values with shift^(n/period)."""
    spec = AIRS[air]
    return [lde(col, BLOWUP, pow(shift, n // spec["period"], P)) for col in spec["periodic"]()]


# ---- FRI ---------------------------------------------------------------------------------------
def fri_fold(values, beta, shift):
    """values = f on shift * <w_M> (natural order).  Returns g on shift^2 * <w_{M/2}> with
    g(x^2) = (f(x) + f(-x))/2 + beta (f(x) - f(-x)) / (2x)."""
    m = len(values)
    w = root_of_unity(m.bit_length() - 1)
    inv2 = pow(2, -1, P)
    half = m // 2
    out = []
    x = shift
    for i in range(half):
        a, b = values[i], values[i + half]
        out.append(((a + b) * inv2 + beta * (a - b) % P * pow(2 * x, -1, P)) % P)
        x = x * w % P
    return out


def poly_degree_bound_check(values, shift, max_deg):
    """True iff the interpolant of `values` on shift*<w_M> has degree <= max_deg."""
    m = len(values)
    coeffs = intt(values, root_of_unity(m.bit_length() - 1))
    return all(c == 0 for c in coeffs[max_deg + 1 :])


# ---- commitments ---------------------------------------------------------------------------------
def commit_rows(columns):
    """Merkle root over rows: leaf = H(...H(H(c0, c1), c2)..., ck) (a single column commits the
    felts themselves), node = H(left, right)."""
    n = len(columns[0])
    leaves = []
    for i in range(n):
        acc = columns[0][i]
        for col in columns[1:]:
            acc = R.pedersen_hash(acc, col[i])
        leaves.append(acc)
    return R.merkle_root(leaves)


# ---- verifier of starkperp.stark.prove (test infrastructure) -------------------------------------
import hashlib as _hashlib


class Transcript:
    """Twin of starkperp.stark.Transcript: chained SHA-256 state, challenges drawn from it."""

    def __init__(self, air, n, shift, seed, public_inputs=()):
        self.state = _hashlib.sha256(b"starkperp/airfri/v2").digest()
        self.absorb("statement:" + air, n, shift, seed, len(public_inputs), *public_inputs)

    def absorb(self, label, *values):
        h = _hashlib.sha256(self.state + label.encode())
        for v in values:
            h.update(int(v).to_bytes(32, "big"))
        self.state = h.digest()

    def challenge(self, label, index=0, modulus=P):
        d = _hashlib.sha256(self.state + label.encode() + int(index).to_bytes(8, "big")).digest()
        return int.from_bytes(d + _hashlib.sha256(d).digest(), "big") % modulus


def _root_from_path(leaf, index, path, hash2):
    node = leaf
    for sib in path:
        node = hash2(sib, node) if index & 1 else hash2(node, sib)
        index >>= 1
    return node


def verify_proof(proof, hash2=R.pedersen_hash, final_log=6, air=None):
    """Checks a proof produced by the GPU prover: Merkle openings, the AIR relation between the
    opened trace rows and the composition column at every queried point, FRI fold consistency down
    to the final layer, and the degree bound of the final layer.  Returns (ok, reason)."""
    n, seed, shift = proof["n"], proof["seed"], proof["shift"]
    spec = AIRS[air or proof.get("air", "pedersen")]
    if (air or proof.get("air")) == "ecdsa":
        # the statement: (z, r, s, Qx, Qy) per signature with the pre-asserts of signature.py:219-241; w, the
        # scalar of the third ladder, is s^-1 mod N (:220) and must pass :226
        pub = proof.get("public_inputs", [])
        if len(pub) != 5 * (n // 1024):
            return False, "public inputs"
        for k in range(n // 1024):
            z, r, sig_s, qx, qy = pub[5 * k : 5 * k + 5]
            if not (1 <= sig_s < R.EC_ORDER and 1 <= r < 2**251 and 0 < z < 2**251 and R.is_point_on_curve(qx, qy)):
                return False, "public inputs"
            if not 1 <= R.inv_mod_curve_size(sig_s) < 2**251:
                return False, "public inputs"
    if (air or proof.get("air")) == "range_check":
        pub = proof.get("public_inputs", [])
        if len(pub) != n // RANGE_CHECK_BITS or not all(0 <= v < 2**RANGE_CHECK_BITS for v in pub):
            return False, "public inputs"
    m = BLOWUP * n
    log_m = m.bit_length() - 1
    root_t, roots, final = proof["trace_root"], proof["layer_roots"], proof["final_layer"]
    n_layers = log_m - final_log
    if len(roots) != n_layers or len(final) != 1 << final_log:
        return False, "shape"
    # the challenges are re-derived from the chained transcript: statement, trace root, then every
    # layer root in order, then the final layer
    air_name = air or proof.get("air", "pedersen")
    tr = Transcript(air_name, n, shift, seed, proof.get("public_inputs", ()))
    tr.absorb("trace_root", root_t)
    alphas = [tr.challenge("alpha", k) for k in range(spec["n_constraints"])]
    betas = []
    for k in range(n_layers):
        tr.absorb("layer_root", roots[k])
        betas.append(tr.challenge("beta", k + 1))
    tr.absorb("final_layer", *final)
    per = periodic_lde(n, shift, air or proof.get("air", "pedersen"))
    w = root_of_unity(log_m)
    zinv = [pow((pow(shift, n, P) * pow(w, n * k, P) - 1) % P, -1, P) for k in range(BLOWUP)]
    # final layer: degree < 3n / 2^n_layers on the domain shift^(2^n_layers) * <w_64>
    fshift = shift
    for _ in range(n_layers):
        fshift = fshift * fshift % P
    if not poly_degree_bound_check(final, fshift, (3 * n >> n_layers) - 1):
        return False, "final layer degree"
    for qi, q in enumerate(proof["queries"]):
        j = tr.challenge("query", qi, modulus=m // 2)
        if q["index"] != j:
            return False, "query index"
        # trace openings: rows j, j+4, j+m/2, j+m/2+4
        want_rows = [j, (j + BLOWUP) % m, j + m // 2, (j + m // 2 + BLOWUP) % m]
        if [t["row"] for t in q["trace"]] != want_rows:
            return False, "trace rows"
        for t in q["trace"]:
            leaf = t["values"][0]
            for v in t["values"][1:]:
                leaf = hash2(leaf, v)
            if _root_from_path(leaf, t["row"], t["path"], hash2) != root_t:
                return False, "trace path"
        # composition at the two layer-0 positions must follow from the trace openings
        for side, pos in enumerate((j, j + m // 2)):
            cur, nxt = q["trace"][2 * side]["values"], q["trace"][2 * side + 1]["values"]
            pv = [tab[pos % (BLOWUP * spec["period"])] for tab in per]
            cv = spec["constraints"](cur, nxt, pv)
            expect = sum(a * c for a, c in zip(alphas, cv)) % P * zinv[pos % BLOWUP] % P
            if q["layers"][0][side]["value"] != expect:
                return False, "composition value"
        # FRI consistency
        jk, s, mk = j, shift, m
        for k in range(n_layers):
            jk %= mk // 2
            a, b = q["layers"][k]
            if (a["pos"], b["pos"]) != (jk, jk + mk // 2):
                return False, "layer positions"
            for o in (a, b):
                if _root_from_path(o["value"], o["pos"], o["path"], hash2) != roots[k]:
                    return False, "layer path"
            wk = root_of_unity(mk.bit_length() - 1)
            x = s * pow(wk, jk, P) % P
            folded = ((a["value"] + b["value"]) * pow(2, -1, P)
                      + betas[k] * (a["value"] - b["value"]) % P * pow(2 * x, -1, P)) % P
            if k + 1 < n_layers:
                nxt_pair = q["layers"][k + 1]
                nxt_val = nxt_pair[0]["value"] if jk < mk // 4 else nxt_pair[1]["value"]
            else:
                nxt_val = final[jk]
            if folded != nxt_val:
                return False, "fold consistency at layer %d" % k
            s = s * s % P
            mk //= 2
    return True, "ok"


# =================================================================================================
# Range-check BUILTIN encoding and the combined builtin trace (build-defined, parity
# unpinned).  The Cairo program is `%builtins output pedersen range_check ecdsa`
# (services/perpetual/cairo/main.cairo:1); its range checks sit on the order path
# (order/order.cairo:53-56: assert_nn / assert_nn_le of the 128-bit fields of the message hash).
#
# rc16 AIR.  A range-checked value is eight 16-bit limbs; every limb is one cell of the column `a`, and the
# statement "every cell lies in [rc_min, rc_max] (a sub-range of [0, 2^16))" is proved as the builtin proves
# it: a second column `s` holds the same cells SORTED, neighbours differ by 0 or 1, s starts at rc_min and ends
# at rc_max, and a permutation argument ties the two columns together - after the two columns are committed a
# challenge z is drawn and a third column p (the SECOND committed phase) accumulates
#     p_i = prod_{j <= i} (z - a_j) / (z - s_j),      p_{n-1} = 1.
# Holes between rc_min and rc_max are filled by the prover with extra range-checked values (the builtin's
# unused cells).  Columns: a (limb), acc (running value: acc = a on limb 0, acc' = 2^16 acc + a' inside a
# value, acc on limb 7 = the 128-bit value), s; phase 2: p.  Rows: 8 per value.
#   C0  first8 (acc - a)                                    / Z_H
#   C1  step8 (acc' - 2^16 acc - a')                        / Z_H
#   C2  (s' - s)(s' - s - 1)              (x - g^(n-1))     / Z_H      every row but the last
#   C3  (p' (z - s') - p (z - a'))        (x - g^(n-1))     / Z_H
#   C4  p (z - s) - (z - a)               / (x - 1)                    first row
#   C5  p - 1                             / (x - g^(n-1))              last row
#   C6  s - rc_min                        / (x - 1)
#   C7  s - rc_max                        / (x - g^(n-1))
# =================================================================================================
RC16_LIMBS = 8
RC16_BITS = 16
N_RC16_CONSTRAINTS = 8


def rc16_fill(values, total_values):
    """values (each < 2^128) -> (padded list of `total_values` values, rc_min, rc_max): the padding values'
    limbs fill the holes of the limb range so that the sorted column can walk it in steps of 0 or 1."""
    assert all(0 <= v < 1 << (RC16_LIMBS * RC16_BITS) for v in values) and values
    limbs = {(v >> (RC16_BITS * k)) & 0xFFFF for v in values for k in range(RC16_LIMBS)}
    lo, hi = min(limbs), max(limbs)
    holes = [x for x in range(lo, hi + 1) if x not in limbs]
    pads = []
    for i in range(0, len(holes), RC16_LIMBS):
        group = holes[i : i + RC16_LIMBS]
        group += [lo] * (RC16_LIMBS - len(group))
        pads.append(sum(l << (RC16_BITS * (RC16_LIMBS - 1 - k)) for k, l in enumerate(group)))
    assert len(values) + len(pads) <= total_values, "the trace is too short to fill the holes of the limb range"
    filler = sum(lo << (RC16_BITS * k) for k in range(RC16_LIMBS))
    return list(values) + pads + [filler] * (total_values - len(values) - len(pads)), lo, hi


def rc16_trace(values):
    """[a, acc, s] for an already padded list of values (rc16_fill)."""
    a, acc = [], []
    for v in values:
        for k in range(RC16_LIMBS):
            run = v >> (RC16_BITS * (RC16_LIMBS - 1 - k))
            acc.append(run)
            a.append(run & 0xFFFF)
    return [a, acc, sorted(a)]


def rc16_product_column(a, s, z):
    p, run = [], 1
    for x, y in zip(a, s):
        run = run * ((z - x) % P) % P * pow((z - y) % P, -1, P) % P
        p.append(run)
    return p


def rc16_periodic_columns():
    return [[1, 0, 0, 0, 0, 0, 0, 0], [1, 1, 1, 1, 1, 1, 1, 0]]


def rc16_constraint_values(cur, nxt, per, pcur, pnxt, z, rc_min, rc_max):
    """[C0 .. C7] numerators (see the table above); cur / nxt = (a, acc, s) at the row and the next one."""
    a, acc, s = cur
    an, accn, sn = nxt
    first8, step8 = per
    d = (sn - s) % P
    return [first8 * (acc - a) % P,
            step8 * (accn - (acc << RC16_BITS) - an) % P,
            d * (d - 1) % P,
            (pnxt * (z - sn) - pcur * (z - an)) % P,
            (pcur * (z - s) - (z - a)) % P,
            (pcur - 1) % P,
            (s - rc_min) % P,
            (s - rc_max) % P]


def rc16_composition_value(alphas, cv, x, zinv, g_last):
    """The rc16 part of the composition column at the point x (zinv = 1 / (x^n - 1))."""
    trans = (alphas[0] * cv[0] + alphas[1] * cv[1] + (alphas[2] * cv[2] + alphas[3] * cv[3]) % P * (x - g_last)) % P
    first = (alphas[4] * cv[4] + alphas[6] * cv[6]) % P * pow((x - 1) % P, -1, P)
    last = (alphas[5] * cv[5] + alphas[7] * cv[7]) % P * pow((x - g_last) % P, -1, P)
    return (trans * zinv + first + last) % P


def rc16_composition_on_coset(cols_lde, p_lde, n, alphas, z, rc_min, rc_max, shift=GEN):
    m = BLOWUP * n
    w = root_of_unity(m.bit_length() - 1)
    g_last = pow(root_of_unity(n.bit_length() - 1), n - 1, P)
    per_lde = [lde(col, BLOWUP, pow(shift, n // 8, P)) for col in rc16_periodic_columns()]
    zinv = [pow((pow(shift, n, P) * pow(w, n * k, P) - 1) % P, -1, P) for k in range(BLOWUP)]
    out, x = [], shift
    for i in range(m):
        j = (i + BLOWUP) % m
        cv = rc16_constraint_values([c[i] for c in cols_lde], [c[j] for c in cols_lde], [t[i % 32] for t in per_lde],
                                    p_lde[i], p_lde[j], z, rc_min, rc_max)
        out.append(rc16_composition_value(alphas, cv, x, zinv[i % BLOWUP], g_last))
        x = x * w % P
    return out


# ---- one trace for the three builtins ------------------------------------------------------------
# Segments side by side at fixed ratios: per 1024 rows two Pedersen hashes (4 columns), one ECDSA verification
# (10 columns) and 128 rc16 cells = 16 range-checked values (3 columns + 1 in the second phase).  The
# composition is the sum of the three AIRs' compositions with independent coefficients.
BUILTIN_SEGMENTS = [("pedersen", 4, N_CONSTRAINTS), ("ecdsa", 10, N_ECDSA_CONSTRAINTS), ("rc16", 3, N_RC16_CONSTRAINTS)]


def builtin_layout(segments):
    """[(air, first column, columns, first alpha, alphas)] for the named segments, in BUILTIN_SEGMENTS order."""
    out, col, al = [], 0, 0
    for air, ncols, nalpha in BUILTIN_SEGMENTS:
        if air in segments:
            out.append((air, col, ncols, al, nalpha))
            col += ncols
            al += nalpha
    return out


def verify_builtins_proof(proof, hash2=R.pedersen_hash, final_log=6):
    """Verifier of starkperp.stark.prove_builtins: two committed phases (the builtin columns, then the
    permutation product of the range-check segment drawn after the challenge z), one composition over all
    segments, FRI.  Returns (ok, reason)."""
    n, seed, shift = proof["n"], proof["seed"], proof["shift"]
    segments = proof["segments"]
    layout = builtin_layout(segments)
    if [s for s, *_ in layout] != list(segments) or not layout:
        return False, "segments"
    n_cols = sum(c for _, _, c, _, _ in layout)
    n_alphas = sum(a for *_, a in layout)
    has_rc = "rc16" in segments
    pub = proof["public_inputs"]
    if has_rc:
        rc_min, rc_max = pub["rc_min"], pub["rc_max"]
        if not (0 <= rc_min <= rc_max < 1 << RC16_BITS) or n % 8:
            return False, "public inputs"
    if "ecdsa" in segments:
        sigs = pub["signatures"]
        if n % 1024 or len(sigs) != n // 1024:
            return False, "public inputs"
        for z_, r_, s_, qx, qy in sigs:
            if not (1 <= s_ < R.EC_ORDER and 1 <= r_ < 2**251 and 0 < z_ < 2**251 and R.is_point_on_curve(qx, qy)):
                return False, "public inputs"
            if not 1 <= R.inv_mod_curve_size(s_) < 2**251:
                return False, "public inputs"
    if "pedersen" in segments and n % 512:
        return False, "public inputs"
    m = BLOWUP * n
    log_m = m.bit_length() - 1
    n_layers = log_m - final_log
    roots, final = proof["layer_roots"], proof["final_layer"]
    if len(roots) != n_layers or len(final) != 1 << final_log:
        return False, "shape"
    flat_pub = [len(segments)] + [BUILTIN_SEGMENTS.index(next(b for b in BUILTIN_SEGMENTS if b[0] == s)) for s in segments]
    if has_rc:
        flat_pub += [rc_min, rc_max]
    if "ecdsa" in segments:
        flat_pub += [v for sig in pub["signatures"] for v in sig]
    tr = Transcript("builtins", n, shift, seed, flat_pub)
    tr.absorb("phase1_root", proof["phase1_root"])
    z = tr.challenge("rc16_z") if has_rc else 0
    if has_rc:
        tr.absorb("phase2_root", proof["phase2_root"])
    alphas = [tr.challenge("alpha", k) for k in range(n_alphas)]
    betas = []
    for k in range(n_layers):
        tr.absorb("layer_root", roots[k])
        betas.append(tr.challenge("beta", k + 1))
    tr.absorb("final_layer", *final)
    w = root_of_unity(log_m)
    g_last = pow(root_of_unity(n.bit_length() - 1), n - 1, P)
    zinv = [pow((pow(shift, n, P) * pow(w, n * k, P) - 1) % P, -1, P) for k in range(BLOWUP)]
    pers = {}
    for air, *_ in layout:
        pers[air] = ([lde(c, BLOWUP, pow(shift, n // 8, P)) for c in rc16_periodic_columns()] if air == "rc16"
                     else periodic_lde(n, shift, air))
    fshift = shift
    for _ in range(n_layers):
        fshift = fshift * fshift % P
    if not poly_degree_bound_check(final, fshift, (3 * n >> n_layers) - 1):
        return False, "final layer degree"
    for qi, q in enumerate(proof["queries"]):
        j = tr.challenge("query", qi, modulus=m // 2)
        if q["index"] != j:
            return False, "query index"
        want_rows = [j, (j + BLOWUP) % m, j + m // 2, (j + m // 2 + BLOWUP) % m]
        if [t["row"] for t in q["phase1"]] != want_rows or (has_rc and [t["row"] for t in q["phase2"]] != want_rows):
            return False, "opened rows"
        for t in q["phase1"]:
            if len(t["values"]) != n_cols:
                return False, "opened rows"
            leaf = t["values"][0]
            for v in t["values"][1:]:
                leaf = hash2(leaf, v)
            if _root_from_path(leaf, t["row"], t["path"], hash2) != proof["phase1_root"]:
                return False, "phase 1 path"
        if has_rc:
            for t in q["phase2"]:
                if _root_from_path(t["value"], t["row"], t["path"], hash2) != proof["phase2_root"]:
                    return False, "phase 2 path"
        for side, pos in enumerate((j, j + m // 2)):
            cur, nxt = q["phase1"][2 * side]["values"], q["phase1"][2 * side + 1]["values"]
            x = shift * pow(w, pos, P) % P
            expect = 0
            for air, c0, nc, a0, na in layout:
                al = alphas[a0 : a0 + na]
                if air == "rc16":
                    pv = [tab[pos % 32] for tab in pers[air]]
                    cv = rc16_constraint_values(cur[c0 : c0 + nc], nxt[c0 : c0 + nc], pv, q["phase2"][2 * side]["value"],
                                                q["phase2"][2 * side + 1]["value"], z, rc_min, rc_max)
                    expect += rc16_composition_value(al, cv, x, zinv[pos % BLOWUP], g_last)
                else:
                    spec = AIRS[air]
                    pv = [tab[pos % (BLOWUP * spec["period"])] for tab in pers[air]]
                    cv = spec["constraints"](cur[c0 : c0 + nc], nxt[c0 : c0 + nc], pv)
                    expect += sum(a * c for a, c in zip(al, cv)) % P * zinv[pos % BLOWUP]
            if q["layers"][0][side]["value"] != expect % P:
                return False, "composition value"
        jk, s, mk = j, shift, m
        for k in range(n_layers):
            jk %= mk // 2
            a, b = q["layers"][k]
            if (a["pos"], b["pos"]) != (jk, jk + mk // 2):
                return False, "layer positions"
            for o in (a, b):
                if _root_from_path(o["value"], o["pos"], o["path"], hash2) != roots[k]:
                    return False, "layer path"
            wk = root_of_unity(mk.bit_length() - 1)
            xk = s * pow(wk, jk, P) % P
            folded = ((a["value"] + b["value"]) * pow(2, -1, P)
                      + betas[k] * (a["value"] - b["value"]) % P * pow(2 * xk, -1, P)) % P
            if k + 1 < n_layers:
                nxt_pair = q["layers"][k + 1]
                nxt_val = nxt_pair[0]["value"] if jk < mk // 4 else nxt_pair[1]["value"]
            else:
                nxt_val = final[jk]
            if folded != nxt_val:
                return False, "fold consistency at layer %d" % k
            s = s * s % P
            mk //= 2
    return True, "ok"

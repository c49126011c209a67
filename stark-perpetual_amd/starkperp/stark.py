"""Prover-side pipeline on device-resident columns: trace -> coset LDE -> commit -> AIR composition
-> commit -> FRI folds with per-layer commits.  Build-defined (the reference has no prover); every
commitment hash is the reference's pedersen_hash.  Columns live in HBM as torch int64 [n, 4]
tensors (four little-endian 64-bit limbs per felt); torch is only the allocator / stream owner."""
from . import _lib
from ._lib import pack_felts

FIELD_PRIME = 2**251 + 17 * 2**192 + 1
FIELD_GEN = 3
BLOWUP_LOG = 2
N_CONSTRAINTS = 11
SHIFT_POINT = (
    0x49EE3EBA8C1600700EE1B87EB599F16716B0B1022947733551FDE4050CA6804,
    0x3CA0CFE4B3BC6DDF346D49D06EA0ED34E621062C0E056C1D0405D266E10268A,
)
EC_GEN = (
    0x1EF15C18599971B7BECED415A40F0C7DEACFD9B0D1819E03D723D8BC943CFCA,
    0x5668060AA49730B7BE4801DF46EC62DE53ECD11ABE43A32873000C36E8DC1F,
)


def _torch():
    import torch
    return torch


def felts_to_tensor(values, device="cuda"):
    torch = _torch()
    import numpy as np
    raw = b"".join(int(v).to_bytes(32, "little") for v in values)
    arr = np.frombuffer(raw, dtype="<i8").reshape(len(values), 4).copy()
    return torch.from_numpy(arr).to(device)


def tensor_to_felts(t):
    raw = t.detach().cpu().contiguous().numpy().astype("<i8").tobytes()
    return [int.from_bytes(raw[32 * i : 32 * i + 32], "little") for i in range(len(raw) // 32)]


def _stream():
    return _torch().cuda.current_stream().cuda_stream


def ntt(col, inverse=False):
    """Natural-order NTT of a [n, 4] column (n a power of two)."""
    torch = _torch()
    lib = _lib.ensure_init()
    n = col.shape[0]
    out = torch.empty_like(col)
    _lib.check(lib.sp_ntt_dev(col.data_ptr(), out.data_ptr(), n.bit_length() - 1, 1 if inverse else 0,
                              _stream()), "sp_ntt_dev")
    return out


def lde(cols, blowup_log=BLOWUP_LOG, shift=FIELD_GEN):
    """cols: [ncols, n, 4] -> [ncols, n << blowup_log, 4] on the coset shift * <w>."""
    torch = _torch()
    lib = _lib.ensure_init()
    ncols, n = cols.shape[0], cols.shape[1]
    out = torch.empty((ncols, n << blowup_log, 4), dtype=torch.int64, device=cols.device)
    _lib.check(lib.sp_lde_dev(cols.data_ptr(), out.data_ptr(), ncols, n.bit_length() - 1, blowup_log,
                              pack_felts([shift]), _stream()), "sp_lde_dev")
    return out


def pedersen_trace(xs, ys):
    """xs, ys: [m, 4] device tensors of hash inputs -> [4, 512 m, 4] trace (s, px, py, lambda)."""
    torch = _torch()
    lib = _lib.ensure_init()
    m = xs.shape[0]
    cols = torch.empty((4, 512 * m, 4), dtype=torch.int64, device=xs.device)
    _lib.check(lib.sp_pedersen_trace_dev(xs.data_ptr(), ys.data_ptr(), m, cols.data_ptr(), _stream()),
               "sp_pedersen_trace_dev")
    return cols


def periodic_columns():
    """The six period-512 columns of the AIR (constant points and selectors) as Python ints; the
    per-bit constant points come from starkperp.signature.CONSTANT_POINTS."""
    from .signature import CONSTANT_POINTS
    cx, cy, step, mid, end, z252 = [], [], [], [], [], []
    for r in range(512):
        block, j = divmod(r, 256)
        c = CONSTANT_POINTS[2 + 252 * block + j] if j < 252 else EC_GEN
        cx.append(c[0])
        cy.append(c[1])
        step.append(0 if j == 255 else 1)
        mid.append(1 if r == 255 else 0)
        end.append(1 if r == 511 else 0)
        z252.append(1 if j == 252 else 0)
    return [cx, cy, step, mid, end, z252]


def ec_ladder_periodic_columns():
    """Period-256 selectors of the EC-ladder AIR: step (rows 0..254), first (row 0), z251."""
    return [[0 if j == 255 else 1 for j in range(256)], [1 if j == 0 else 0 for j in range(256)],
            [1 if j == 251 else 0 for j in range(256)]]


def ecdsa_periodic_columns():
    """Period-1024 tables of the ECDSA-verification AIR (oracle/stark_ref.py ecdsa_periodic_columns is the
    definition): step, first, start_y, z251, gbase, oncurve, carry_load, carry_hold, addb, rload, rhold, fin."""
    def rows(fn):
        return [1 if fn(i) else 0 for i in range(1024)]
    ladder = lambda i: i < 768
    start_y = [0] * 1024
    start_y[0], start_y[256], start_y[512] = FIELD_PRIME - SHIFT_POINT[1], SHIFT_POINT[1], SHIFT_POINT[1]
    return [rows(lambda i: ladder(i) and i % 256 != 255), rows(lambda i: ladder(i) and i % 256 == 0), start_y,
            rows(lambda i: ladder(i) and i % 256 == 251), rows(lambda i: i == 0), rows(lambda i: i == 256),
            rows(lambda i: i == 255), rows(lambda i: 256 <= i < 511), rows(lambda i: i == 511),
            rows(lambda i: i == 256), rows(lambda i: 256 <= i < 767), rows(lambda i: i == 767)]


def range_check_periodic_columns():
    """step (rows 0..126 of every 128), last (row 127): oracle/stark_ref.py range_check_periodic_columns."""
    return [[1] * 127 + [0], [0] * 127 + [1]]


N_RANGE_CHECK_CONSTRAINTS = 2
RANGE_CHECK_BITS = 128
N_EC_LADDER_CONSTRAINTS = 12
N_ECDSA_CONSTRAINTS = 26
EC_ORDER = 0x800000000000010FFFFFFFFFFFFFFFFB781126DCAE7B2321E66A241ADC64D2F
AIRS = {
    "ecdsa": {"n_cols": 10, "period": 1024, "n_constraints": N_ECDSA_CONSTRAINTS, "periodic": ecdsa_periodic_columns,
              "eval": "sp_air_eval_ecdsa_dev"},
    "pedersen": {"n_cols": 4, "period": 512, "n_constraints": N_CONSTRAINTS, "periodic": periodic_columns,
                 "eval": "sp_air_eval_dev"},
    "ec_ladder": {"n_cols": 7, "period": 256, "n_constraints": N_EC_LADDER_CONSTRAINTS,
                  "periodic": ec_ladder_periodic_columns, "eval": "sp_air_eval_ec_ladder_dev"},
    "range_check": {"n_cols": 1, "period": 128, "n_constraints": N_RANGE_CHECK_CONSTRAINTS,
                    "periodic": range_check_periodic_columns, "eval": "sp_air_eval_range_check_dev"},
}


def rc16_periodic_columns():
    """first8 (limb 0 of a value), step8 (the next row belongs to the same value): oracle/stark_ref.py rc16."""
    return [[1, 0, 0, 0, 0, 0, 0, 0], [1, 1, 1, 1, 1, 1, 1, 0]]


N_RC16_CONSTRAINTS = 8
RC16_LIMBS = 8
AIRS["rc16"] = {"n_cols": 3, "period": 8, "n_constraints": N_RC16_CONSTRAINTS, "periodic": rc16_periodic_columns,
                "eval": None}  # evaluated by air_eval_rc16 (it needs the second-phase column and the challenge z)
BUILTIN_SEGMENTS = ["pedersen", "ecdsa", "rc16"]  # column order of a combined builtin trace


def periodic_lde(n, shift=FIELD_GEN, device="cuda", air="pedersen"):
    """[k, 4 * period, 4]: the periodic columns on the LDE coset (they repeat with period 4 * period)."""
    torch = _torch()
    spec = AIRS[air]
    cols = torch.stack([felts_to_tensor(c, device) for c in spec["periodic"]()])
    return lde(cols, BLOWUP_LOG, pow(shift, n // spec["period"], FIELD_PRIME))


def air_eval(trace_lde, per_lde, n, alphas, shift=FIELD_GEN, air="pedersen"):
    torch = _torch()
    lib = _lib.ensure_init()
    spec = AIRS[air]
    assert len(alphas) == spec["n_constraints"] and trace_lde.shape[0] == spec["n_cols"]
    assert trace_lde.shape[1] == 4 * n
    out = torch.empty((4 * n, 4), dtype=torch.int64, device=trace_lde.device)
    _lib.check(getattr(lib, spec["eval"])(trace_lde.data_ptr(), per_lde.data_ptr(), n.bit_length() - 1,
                                          pack_felts(alphas), pack_felts([shift]), out.data_ptr(), _stream()),
               spec["eval"])
    return out


def ec_ladder_trace(ms, qxs, qys):
    """ms, qxs, qys: [k, 4] device tensors (scalars 0 < m < 2^251, affine base points) ->
    [7, 256 k, 4] witness of k mimic_ec_mult_air ladders (m, px, py, qx, qy, la, ld)."""
    torch = _torch()
    lib = _lib.ensure_init()
    k = ms.shape[0]
    cols = torch.empty((7, 256 * k, 4), dtype=torch.int64, device=ms.device)
    _lib.check(lib.sp_ec_ladder_trace_dev(ms.data_ptr(), qxs.data_ptr(), qys.data_ptr(), k, cols.data_ptr(),
                                          _stream()), "sp_ec_ladder_trace_dev")
    return cols


def ecdsa_trace(zs, rs, ws, qxs, qys):
    """zs, rs, ws, qxs, qys: [k, 4] device tensors (message hashes, r, w = s^-1 mod N, public key points) of
    signatures verify() accepts -> [10, 1024 k, 4] witness of the ECDSA-verification AIR."""
    torch = _torch()
    lib = _lib.ensure_init()
    k = zs.shape[0]
    cols = torch.empty((10, 1024 * k, 4), dtype=torch.int64, device=zs.device)
    _lib.check(lib.sp_ecdsa_trace_dev(zs.data_ptr(), rs.data_ptr(), ws.data_ptr(), qxs.data_ptr(), qys.data_ptr(), k,
                                      cols.data_ptr(), _stream()), "sp_ecdsa_trace_dev")
    return cols


def range_check_trace(values):
    """values: [k, 4] device tensor -> [1, 128 k, 4] witness of the range-check AIR (v_i = value >> i)."""
    torch = _torch()
    lib = _lib.ensure_init()
    k = values.shape[0]
    col = torch.empty((1, RANGE_CHECK_BITS * k, 4), dtype=torch.int64, device=values.device)
    _lib.check(lib.sp_range_check_trace_dev(values.data_ptr(), k, col.data_ptr(), _stream()),
               "sp_range_check_trace_dev")
    return col


def fri_fold(layer, beta, shift):
    torch = _torch()
    lib = _lib.ensure_init()
    m = layer.shape[0]
    out = torch.empty((m // 2, 4), dtype=torch.int64, device=layer.device)
    _lib.check(lib.sp_fri_fold_dev(layer.data_ptr(), out.data_ptr(), m.bit_length() - 1,
                                   pack_felts([beta]), pack_felts([shift]), _stream()),
               "sp_fri_fold_dev")
    return out


def commit_rows(cols):
    """Pedersen-Merkle root over rows of `cols` [ncols, n, 4]: leaf = left-fold hash of the row
    (a single column commits the felts themselves).  Returns (root_int, levels tensor)."""
    torch = _torch()
    lib = _lib.ensure_init()
    ncols, n = cols.shape[0], cols.shape[1]
    levels = torch.empty((2 * n - 1, 4), dtype=torch.int64, device=cols.device)
    _lib.check(lib.sp_commit_rows_dev(cols.contiguous().data_ptr(), n, ncols, levels.data_ptr(), None, _stream()),
               "sp_commit_rows_dev")
    return levels


def root_of(levels):
    return tensor_to_felts(levels[-1:])[0]


def prove_commitments(xs, ys, alphas, betas, final_log=6, shift=FIELD_GEN):
    """The AIR+FRI commit job of BASELINE.json configs[3]: returns the list of commitment roots
    [trace, composition, fri_1, ..., fri_k] and the final FRI layer (Python ints)."""
    n = 512 * xs.shape[0]
    trace = pedersen_trace(xs, ys)
    trace_lde = lde(trace)
    roots = [commit_rows(trace_lde)]
    per = periodic_lde(n, shift, xs.device)
    comp = air_eval(trace_lde, per, n, alphas, shift)
    roots.append(commit_rows(comp.unsqueeze(0)))
    layer, s, k = comp, shift, 0
    while layer.shape[0] > (1 << final_log):
        layer = fri_fold(layer, betas[k], s)
        s = s * s % FIELD_PRIME
        k += 1
        if layer.shape[0] > (1 << final_log):
            roots.append(commit_rows(layer.unsqueeze(0)))
    return [root_of(r) for r in roots], tensor_to_felts(layer)


# ---- a checkable proof: Fiat-Shamir transcript, query openings -----------------------------------
# Build-defined like the rest of this module (the reference has no prover).  The transcript hash
# is SHA-256 (host); every commitment / opening hash is the reference's pedersen_hash on the GPU.
import hashlib as _hashlib


class Transcript:
    """Running SHA-256 Fiat-Shamir transcript: every challenge depends on the statement and on EVERY
    commitment absorbed before it (state <- SHA256(state || label || values))."""

    def __init__(self, air: str, n: int, shift: int, seed: int, public_inputs=()):
        self.state = _hashlib.sha256(b"starkperp/airfri/v2").digest()
        self.absorb("statement:" + air, n, shift, seed, len(public_inputs), *public_inputs)

    def absorb(self, label: str, *values: int):
        h = _hashlib.sha256(self.state + label.encode())
        for v in values:
            h.update(int(v).to_bytes(32, "big"))
        self.state = h.digest()

    def challenge(self, label: str, index: int = 0, modulus: int = FIELD_PRIME) -> int:
        d = _hashlib.sha256(self.state + label.encode() + int(index).to_bytes(8, "big")).digest()
        return int.from_bytes(d + _hashlib.sha256(d).digest(), "big") % modulus


def _path_indices(n_leaves: int, idx: int):
    """Flat indices (into a leaves-first levels buffer) of the siblings along the path of leaf idx."""
    out, off, width, i = [], 0, n_leaves, idx
    while width > 1:
        out.append(off + (i ^ 1))
        off += width
        width >>= 1
        i >>= 1
    return out


def _gather_felts(t, indices):
    torch = _torch()
    idx = torch.tensor(indices, dtype=torch.int64, device=t.device)
    return tensor_to_felts(t.index_select(0, idx))


def prove(xs, ys, n_queries: int = 8, seed: int = 0, final_log: int = 6, shift: int = FIELD_GEN):
    """Benchmark-grade argument (NOT a sound proof system) that the trace of the hashes (xs[i], ys[i])
    satisfies the Pedersen-step AIR: see prove_trace for what is and is not bound."""
    return prove_trace(pedersen_trace(xs, ys), "pedersen", n_queries, seed, final_log, shift)


def prove_ec_ladders(ms, qxs, qys, n_queries: int = 8, seed: int = 0, final_log: int = 6,
                     shift: int = FIELD_GEN):
    """Benchmark-grade argument (NOT a sound proof system) that k scalar multiplications
    m * Q + SHIFT_POINT were carried out step by step as mimic_ec_mult_air does (the EC-ladder AIR)."""
    return prove_trace(ec_ladder_trace(ms, qxs, qys), "ec_ladder", n_queries, seed, final_log, shift)


def prove_ecdsa(msg_hashes, rs, ss, public_keys, n_queries: int = 8, seed: int = 0, final_log: int = 6,
                shift: int = FIELD_GEN):
    """Benchmark-grade argument (NOT a sound proof system) that every (z, r, s, Q) passes the reference's
    verify (signature.py:217-260: the three ladders, the two additions, x == r).  Python ints in; the
    count must be a power of two.  w = s^-1 mod N is computed here as verify() does (:220) and the public
    inputs (z, r, s, Qx, Qy per signature) are absorbed into the transcript; the CPU verifier
    (oracle/stark_ref.verify_proof) re-derives w from s."""
    dev = "cuda"
    ws = [pow(s, -1, EC_ORDER) for s in ss]
    for z, r, w in zip(msg_hashes, rs, ws):
        assert 0 < z < 2**251 and 1 <= r < 2**251 and 1 <= w < 2**251
    trace = ecdsa_trace(*(felts_to_tensor(v, dev) for v in (msg_hashes, rs, ws, [q[0] for q in public_keys],
                                                          [q[1] for q in public_keys])))
    public = [v for z, r, s, q in zip(msg_hashes, rs, ss, public_keys) for v in (z, r, s, q[0], q[1])]
    return prove_trace(trace, "ecdsa", n_queries, seed, final_log, shift, public)


def prove_range_checks(values, n_queries: int = 8, seed: int = 0, final_log: int = 6, shift: int = FIELD_GEN):
    """Benchmark-grade argument (NOT a sound proof system) that every value lies in [0, 2^128) - the bound
    the Cairo range-check builtin gives the amounts, ids, nonces and timestamps of the exchange messages.
    Python ints in; the count must be a power of two.  The values are absorbed as public inputs."""
    values = list(values)
    assert values and len(values) & (len(values) - 1) == 0
    for v in values:
        assert 0 <= v < FIELD_PRIME
    return prove_trace(range_check_trace(felts_to_tensor(values, "cuda")), "range_check", n_queries, seed, final_log,
                       shift, values)


def prove_trace(trace, air: str, n_queries: int = 8, seed: int = 0, final_log: int = 6,
                shift: int = FIELD_GEN, public_inputs=()):
    """Commitments to the trace LDE, the composition column and every FRI layer, the final layer in
    the clear, and for each query the openings a verifier needs (oracle/stark_ref.verify_proof).

    What this is: the prover-side workload of BASELINE.json configs[3] made checkable end to end.  The
    Fiat-Shamir transcript is chained (statement, then every commitment in order; alpha, every beta and
    the query positions are drawn from it).  What it is not: a sound, complete STARK - no boundary
    constraint binds the hash inputs / outputs (or `public_inputs`, which are only absorbed), there is
    no zero knowledge, and the default 8 queries at blowup 4 give about 16 bits.  The reference has no
    prover; this path is build-defined and benchmark-grade."""
    P = FIELD_PRIME
    spec = AIRS[air]
    n = trace.shape[1]
    M = n << BLOWUP_LOG
    tr = Transcript(air, n, shift, seed, public_inputs)
    trace_lde = lde(trace)
    lv_trace = commit_rows(trace_lde)
    root_t = root_of(lv_trace)
    tr.absorb("trace_root", root_t)
    alphas = [tr.challenge("alpha", k) for k in range(spec["n_constraints"])]
    per = periodic_lde(n, shift, trace.device, air)
    comp = air_eval(trace_lde, per, n, alphas, shift, air)
    layers, level_bufs, roots = [comp], [], []
    s = shift
    while True:
        cur = layers[-1]
        lv = commit_rows(cur.unsqueeze(0))
        level_bufs.append(lv)
        roots.append(root_of(lv))
        tr.absorb("layer_root", roots[-1])
        beta = tr.challenge("beta", len(roots))
        nxt = fri_fold(cur, beta, s)
        s = s * s % P
        layers.append(nxt)
        if nxt.shape[0] <= (1 << final_log):
            break
    final = tensor_to_felts(layers[-1])
    tr.absorb("final_layer", *final)
    queries = []
    for q in range(n_queries):
        j = tr.challenge("query", q, modulus=M // 2)
        entry = {"index": j, "trace": [], "layers": []}
        for pos in (j, j + M // 2):
            for row in (pos, (pos + (1 << BLOWUP_LOG)) % M):
                vals = [tensor_to_felts(trace_lde[c][row : row + 1])[0] for c in range(spec["n_cols"])]
                entry["trace"].append({"row": row, "values": vals,
                                       "path": _gather_felts(lv_trace, _path_indices(M, row))})
        jk = j
        for k, (layer, lv) in enumerate(zip(layers[:-1], level_bufs)):
            mk = layer.shape[0]
            jk %= mk // 2
            pair = []
            for pos in (jk, jk + mk // 2):
                pair.append({"pos": pos, "value": tensor_to_felts(layer[pos : pos + 1])[0],
                             "path": _gather_felts(lv, _path_indices(mk, pos))})
            entry["layers"].append(pair)
        queries.append(entry)
    return {"n": n, "air": air, "seed": seed, "shift": shift, "public_inputs": list(public_inputs),
            "trace_root": root_t, "layer_roots": roots, "final_layer": final, "queries": queries}


# ---- the range-check builtin's encoding and ONE trace for the three builtins --------------------
# Definition: oracle/stark_ref.py ("rc16", BUILTIN_SEGMENTS, verify_builtins_proof).  The Cairo program is
# `%builtins output pedersen range_check ecdsa` (services/perpetual/cairo/main.cairo:1); no VM is needed for the
# builtin segments - the instance lists (hash inputs, signatures, range-checked values) are the input.
def rc16_fill(values, total_values):
    """values (each < 2^128) -> (padded list of total_values values, rc_min, rc_max); the padding values' limbs
    fill the holes of the limb range (the builtin's unused cells)."""
    limbs = set()
    for v in values:
        assert 0 <= v < 1 << 128
        for k in range(RC16_LIMBS):
            limbs.add((v >> (16 * k)) & 0xFFFF)
    lo, hi = min(limbs), max(limbs)
    holes = [x for x in range(lo, hi + 1) if x not in limbs]
    pads = []
    for i in range(0, len(holes), RC16_LIMBS):
        group = holes[i : i + RC16_LIMBS]
        group += [lo] * (RC16_LIMBS - len(group))
        pads.append(sum(l << (16 * (RC16_LIMBS - 1 - k)) for k, l in enumerate(group)))
    if len(values) + len(pads) > total_values:
        raise ValueError("the trace is too short to fill the %d holes of the limb range" % len(holes))
    filler = sum(lo << (16 * k) for k in range(RC16_LIMBS))
    return list(values) + pads + [filler] * (total_values - len(values) - len(pads)), lo, hi


def rc16_columns(values):
    """values: [k, 4] device tensor (already padded, rc16_fill) -> [3, 8 k, 4]: a (limb), acc (running value), s
    (the limb column sorted)."""
    torch = _torch()
    lib = _lib.ensure_init()
    k = values.shape[0]
    cols = torch.zeros((3, RC16_LIMBS * k, 4), dtype=torch.int64, device=values.device)
    _lib.check(lib.sp_rc16_trace_dev(values.data_ptr(), k, cols.data_ptr(), _stream()), "sp_rc16_trace_dev")
    cols[2, :, 0] = torch.sort(cols[0, :, 0]).values  # limbs are < 2^16: the low word is the felt
    return cols


def rc16_product(a, s, z):
    """The second-phase column p_i = prod_{j <= i} (z - a_j) / (z - s_j) (a, s: [n, 4] device tensors)."""
    torch = _torch()
    lib = _lib.ensure_init()
    n = a.shape[0]
    p = torch.empty((n, 4), dtype=torch.int64, device=a.device)
    _lib.check(lib.sp_rc16_product_dev(a.contiguous().data_ptr(), s.contiguous().data_ptr(), n, pack_felts([z]),
                                       p.data_ptr(), _stream()), "sp_rc16_product_dev")
    return p


def air_eval_rc16(cols_lde, p_lde, per_lde, n, alphas, z, rc_min, rc_max, shift=FIELD_GEN):
    torch = _torch()
    lib = _lib.ensure_init()
    assert cols_lde.shape[0] == 3 and cols_lde.shape[1] == 4 * n and len(alphas) == N_RC16_CONSTRAINTS
    out = torch.empty((4 * n, 4), dtype=torch.int64, device=cols_lde.device)
    _lib.check(lib.sp_air_eval_rc16_dev(cols_lde.data_ptr(), p_lde.data_ptr(), per_lde.data_ptr(), n.bit_length() - 1,
                                        pack_felts(alphas), pack_felts([shift]), pack_felts([z]), rc_min, rc_max,
                                        out.data_ptr(), _stream()), "sp_air_eval_rc16_dev")
    return out


def felt_add(a, b):
    torch = _torch()
    lib = _lib.ensure_init()
    out = torch.empty_like(a)
    _lib.check(lib.sp_felt_add_dev(a.data_ptr(), b.data_ptr(), out.data_ptr(), a.shape[0], _stream()), "sp_felt_add_dev")
    return out


def _cycle(items, count):
    return [items[i % len(items)] for i in range(count)]


def prove_builtins(hash_inputs=None, signatures=None, rc_values=None, n_queries: int = 8, seed: int = 0,
                   final_log: int = 6, shift: int = FIELD_GEN, log_rows: int = None):
    """ONE trace, ONE composition, ONE proof for the builtin usage of a batch: Pedersen hashes
    (hash_inputs: [(x, y)]), ECDSA verifications (signatures: [(z, r, s, (Qx, Qy))]) and range checks (rc_values:
    ints < 2^128) laid side by side at fixed ratios - per 1024 rows two hashes, one verification and 128 rc16
    cells (16 values).  The trace length is the power of two the largest segment needs; the other segments repeat
    their instances (unused builtin cells).  Two committed phases: the builtin columns, then - after the
    challenge z - the permutation product of the range-check segment.  Benchmark-grade like prove_trace (no
    boundary constraint ties the instances to the public inputs except rc_min / rc_max; ~2 bits per query).
    Verifier: oracle/stark_ref.verify_builtins_proof."""
    torch = _torch()
    dev = "cuda"
    P = FIELD_PRIME
    segments = [name for name, arg in zip(BUILTIN_SEGMENTS, (hash_inputs, signatures, rc_values)) if arg]
    assert segments, "no builtin instance given"
    need = 8
    if hash_inputs:
        need = max(need, 512 * len(hash_inputs))
    if signatures:
        need = max(need, 1024 * len(signatures))
    rc_need = 0
    if rc_values:
        limbs = {(v >> (16 * k)) & 0xFFFF for v in rc_values for k in range(RC16_LIMBS)}
        rc_need = len(rc_values) + -(-(max(limbs) - min(limbs) + 1 - len(limbs)) // RC16_LIMBS)
        need = max(need, RC16_LIMBS * rc_need)
    n = 1 << max((need - 1).bit_length(), log_rows or 0)
    M = n << BLOWUP_LOG
    groups, public, flat_pub = [], {}, [len(segments)] + [BUILTIN_SEGMENTS.index(s) for s in segments]
    rc_min = rc_max = 0
    if rc_values:
        padded, rc_min, rc_max = rc16_fill(list(rc_values), n // RC16_LIMBS)
        public.update({"rc_min": rc_min, "rc_max": rc_max})
        flat_pub += [rc_min, rc_max]
    if signatures:
        sigs = _cycle(list(signatures), n // 1024)
        ws = [pow(s_, -1, EC_ORDER) for _, _, s_, _ in sigs]
        for (z_, r_, _, _), w_ in zip(sigs, ws):
            assert 0 < z_ < 2**251 and 1 <= r_ < 2**251 and 1 <= w_ < 2**251
        public["signatures"] = [[z_, r_, s_, q[0], q[1]] for z_, r_, s_, q in sigs]
        flat_pub += [v for sig in public["signatures"] for v in sig]
    if hash_inputs:
        hs = _cycle(list(hash_inputs), n // 512)
        groups.append(pedersen_trace(felts_to_tensor([x for x, _ in hs], dev), felts_to_tensor([y for _, y in hs], dev)))
    if signatures:
        groups.append(ecdsa_trace(*(felts_to_tensor(v, dev) for v in (
            [z_ for z_, _, _, _ in sigs], [r_ for _, r_, _, _ in sigs], ws, [q[0] for *_, q in sigs],
            [q[1] for *_, q in sigs]))))
    if rc_values:
        groups.append(rc16_columns(felts_to_tensor(padded, dev)))
    trace = torch.cat(groups)
    del groups
    n_cols = trace.shape[0]
    tr = Transcript("builtins", n, shift, seed, flat_pub)
    # ---- phase 1: the builtin columns ----
    trace_lde = lde(trace)
    lv1 = commit_rows(trace_lde)
    root1 = root_of(lv1)
    tr.absorb("phase1_root", root1)
    # ---- phase 2: the permutation product of the range-check segment, after its challenge ----
    z, p_lde, lv2, root2 = 0, None, None, None
    if rc_values:
        z = tr.challenge("rc16_z")
        p = rc16_product(trace[n_cols - 3], trace[n_cols - 1], z)
        p_lde = lde(p.unsqueeze(0))[0]
        lv2 = commit_rows(p_lde.unsqueeze(0))
        root2 = root_of(lv2)
        tr.absorb("phase2_root", root2)
    del trace
    n_alphas = sum(AIRS[s]["n_constraints"] for s in segments)
    alphas = [tr.challenge("alpha", k) for k in range(n_alphas)]
    comp, c0, a0 = None, 0, 0
    for seg in segments:
        spec = AIRS[seg]
        sl = trace_lde[c0 : c0 + spec["n_cols"]]
        al = alphas[a0 : a0 + spec["n_constraints"]]
        per = periodic_lde(n, shift, dev, seg)
        part = (air_eval_rc16(sl, p_lde, per, n, al, z, rc_min, rc_max, shift) if seg == "rc16"
                else air_eval(sl, per, n, al, shift, seg))
        comp = part if comp is None else felt_add(comp, part)
        c0 += spec["n_cols"]
        a0 += spec["n_constraints"]
    layers, level_bufs, roots = [comp], [], []
    s = shift
    while True:
        cur = layers[-1]
        lv = commit_rows(cur.unsqueeze(0))
        level_bufs.append(lv)
        roots.append(root_of(lv))
        tr.absorb("layer_root", roots[-1])
        beta = tr.challenge("beta", len(roots))
        nxt = fri_fold(cur, beta, s)
        s = s * s % P
        layers.append(nxt)
        if nxt.shape[0] <= (1 << final_log):
            break
    final = tensor_to_felts(layers[-1])
    tr.absorb("final_layer", *final)
    queries = []
    for q in range(n_queries):
        j = tr.challenge("query", q, modulus=M // 2)
        entry = {"index": j, "phase1": [], "phase2": [], "layers": []}
        for pos in (j, j + M // 2):
            for row in (pos, (pos + (1 << BLOWUP_LOG)) % M):
                entry["phase1"].append({"row": row, "values": tensor_to_felts(trace_lde[:, row]),
                                        "path": _gather_felts(lv1, _path_indices(M, row))})
                if rc_values:
                    entry["phase2"].append({"row": row, "value": tensor_to_felts(p_lde[row : row + 1])[0],
                                            "path": _gather_felts(lv2, _path_indices(M, row))})
        jk = j
        for k, (layer, lv) in enumerate(zip(layers[:-1], level_bufs)):
            mk = layer.shape[0]
            jk %= mk // 2
            pair = []
            for pos in (jk, jk + mk // 2):
                pair.append({"pos": pos, "value": tensor_to_felts(layer[pos : pos + 1])[0],
                             "path": _gather_felts(lv, _path_indices(mk, pos))})
            entry["layers"].append(pair)
        queries.append(entry)
    return {"n": n, "air": "builtins", "segments": segments, "seed": seed, "shift": shift, "public_inputs": public,
            "phase1_root": root1, "phase2_root": root2, "layer_roots": roots, "final_layer": final, "queries": queries}

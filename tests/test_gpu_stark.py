"""GPU parity of the prover-side kernels (NTT / LDE / AIR / FRI / commit) against
oracle/stark_ref.py at oracle-sized inputs, plus size-independent properties at 2^20.
The reference has no prover: "parity unpinned" (see the oracle header)."""
import random

import pytest

from oracle import ref_py as R
from oracle import stark_ref as S

pytestmark = pytest.mark.gpu
P = S.P


@pytest.fixture(scope="module")
def stark():
    from starkperp import stark as st
    return st


def test_ntt_matches_oracle(stark):
    rng = random.Random(1)
    for log_n in (0, 1, 2, 5, 10, 11, 12, 13):
        n = 1 << log_n
        c = [rng.randrange(P) for _ in range(n)]
        w = S.root_of_unity(log_n)
        t = stark.felts_to_tensor(c)
        assert stark.tensor_to_felts(stark.ntt(t)) == S.ntt(c, w), log_n
        assert stark.tensor_to_felts(stark.ntt(stark.ntt(t), inverse=True)) == c


def test_ntt_roundtrip_and_linearity_large(stark):
    import torch
    rng = random.Random(2)
    log_n = 21
    n = 1 << log_n
    g = torch.Generator().manual_seed(5)
    a = torch.randint(0, 2**62, (n, 4), dtype=torch.int64, generator=g)
    a[:, 3] &= (1 << 58) - 1
    a = a.cuda()
    fa = stark.ntt(a)
    assert torch.equal(stark.ntt(fa, inverse=True), a)
    # spot values against the definition: f(w^i) for a sparse polynomial
    coeffs = {0: 5, 1: 7, 12345: 11, n - 1: 13}
    dense = [0] * n
    for k, v in coeffs.items():
        dense[k] = v
    ev = stark.tensor_to_felts(stark.ntt(stark.felts_to_tensor(dense)))
    w = S.root_of_unity(log_n)
    for i in (0, 1, 2, 777, n // 2, n - 1):
        x = pow(w, i, P)
        assert ev[i] == sum(v * pow(x, k, P) for k, v in coeffs.items()) % P


def test_ntt_and_lde_at_the_maximum_size(stark):
    """log_n = 26 (the C-ABI limit; 2^24 trace rows x blowup 4 = BASELINE.json configs[4] on one GPU):
    round trip on random data, and a sparse polynomial checked against its definition both through
    the plain NTT and through the coset LDE 2^24 -> 2^26."""
    import torch
    log_n = 26
    n = 1 << log_n
    g = torch.Generator().manual_seed(9)
    a = torch.randint(0, 2**62, (n, 4), dtype=torch.int64, generator=g)
    a[:, 3] &= (1 << 58) - 1
    a = a.cuda()
    fa = stark.ntt(a)
    assert torch.equal(stark.ntt(fa, inverse=True), a)
    del a, fa
    coeffs = {0: 5, 1: 7, 12345: 11, (1 << 24) - 1: 13}

    def sparse(size):
        t = torch.zeros((size, 4), dtype=torch.int64, device="cuda")
        for k, v in coeffs.items():
            t[k, 0] = v
        return t

    def f(x):
        return sum(v * pow(x, k, P) for k, v in coeffs.items()) % P

    ev = stark.ntt(sparse(n))
    w = S.root_of_unity(log_n)
    spots = [0, 1, 2, 777, n // 2, n - 1, 0x2345678]
    got = stark.tensor_to_felts(ev[spots])
    assert got == [f(pow(w, i, P)) for i in spots]
    del ev
    on_trace_domain = stark.ntt(sparse(1 << 24))  # f on <w_{2^24}>, degree < 2^24
    ext = stark.lde(on_trace_domain.unsqueeze(0))[0]
    assert ext.shape[0] == n
    got = stark.tensor_to_felts(ext[spots])
    assert got == [f(stark.FIELD_GEN * pow(w, i, P) % P) for i in spots]


def test_ntt_and_lde_of_sparse_polynomials_at_every_size(stark):
    """Every pass plan of the register-blocked NTT (local only, one and two strided passes; radix-8 / 4 / 2 groups;
    the unit-twiddle stage) at log_n = 2 .. 22: forward transform and coset LDE (x2, x4) of a sparse polynomial
    against its definition, and the inverse transform back to the coefficients.  Sparse polynomials matter: their
    evaluations contain "minus small" values whose limbs are all 2^29 - 1 - four of those in one lazy sum sit at
    the edge of an int32 limb, which random data never does (a value reduction on such a sum once wrapped limb 6)."""
    import torch
    for log_n in range(2, 23):
        n = 1 << log_n
        coeffs = {0: 5, 1: 7, 57 % n: 11, n - 1: 13, n // 2 + 1: 17}

        def sparse(size):
            t = torch.zeros((size, 4), dtype=torch.int64, device="cuda")
            for k, v in coeffs.items():
                t[k, 0] += v
            return t

        merged = {}
        for k, v in coeffs.items():
            merged[k] = merged.get(k, 0) + v

        def f(x):
            return sum(v * pow(x, k, P) for k, v in merged.items()) % P

        w = S.root_of_unity(log_n)
        spots = sorted({0, 1, 2, 777 % n, n // 2, n - 1, 0x2345678 % n, n // 3})
        ev = stark.ntt(sparse(n))
        assert stark.tensor_to_felts(ev[spots]) == [f(pow(w, i, P)) for i in spots], log_n
        assert torch.equal(stark.ntt(ev, inverse=True), sparse(n)), log_n
        for bl in (1, 2):
            m = n << bl
            wm = S.root_of_unity(log_n + bl)
            ext = stark.lde(ev.unsqueeze(0), blowup_log=bl)[0]
            sp = sorted({0, 1, 2, 3, 777 % m, m // 2, m - 1, 0x2345678 % m, m // 3})
            assert stark.tensor_to_felts(ext[sp]) == [f(stark.FIELD_GEN * pow(wm, i, P) % P) for i in sp], (log_n, bl)


def test_lde_matches_oracle(stark):
    import torch
    rng = random.Random(3)
    for log_n in (4, 9, 11):
        n = 1 << log_n
        cols = [[rng.randrange(P) for _ in range(n)] for _ in range(2)]
        t = torch.stack([stark.felts_to_tensor(c) for c in cols])
        got = stark.lde(t)
        for c, g in zip(cols, got):
            assert stark.tensor_to_felts(g) == S.lde(c), log_n


def test_trace_air_fri_match_oracle(stark):
    import torch
    rng = random.Random(4)
    inputs = [(rng.randrange(P), rng.randrange(P)) for _ in range(2)]
    xs = stark.felts_to_tensor([a for a, _ in inputs])
    ys = stark.felts_to_tensor([b for _, b in inputs])
    n = 1024
    trace = stark.pedersen_trace(xs, ys)
    exp_cols = S.pedersen_trace(inputs)
    for g, e in zip(trace, exp_cols):
        assert stark.tensor_to_felts(g) == e
    assert exp_cols[1][511] == R.pedersen_hash(*inputs[0])
    trace_lde = stark.lde(trace)
    per = stark.periodic_lde(n)
    exp_per = S.periodic_lde(n)
    for g, e in zip(per, exp_per):
        assert stark.tensor_to_felts(g) == e
    alphas = [rng.randrange(P) for _ in range(S.N_CONSTRAINTS)]
    comp = stark.air_eval(trace_lde, per, n, alphas)
    exp_comp = S.composition_on_coset([S.lde(c) for c in exp_cols], exp_per, n, alphas)
    assert stark.tensor_to_felts(comp) == exp_comp
    assert S.poly_degree_bound_check(exp_comp, S.GEN, 3 * n - 1)
    layer, exp_layer, shift = comp, exp_comp, S.GEN
    while layer.shape[0] > 64:
        beta = rng.randrange(P)
        layer = stark.fri_fold(layer, beta, shift)
        exp_layer = S.fri_fold(exp_layer, beta, shift)
        shift = shift * shift % P
        assert stark.tensor_to_felts(layer) == exp_layer


def test_prover_kernels_on_extreme_limb_patterns(stark):
    """NTT / LDE / composition / fold on inputs made of extreme limb patterns (all-ones limbs, p - small, powers of
    two at the limb boundaries) equal the oracle: the kernels reduce lazily, and the cases that overflow a lazy
    reduction first are exactly the ones seeded random data never produces."""
    import torch
    rng = random.Random(44)
    import workloads as wl
    ext = wl.extreme_felts()
    pick = lambda k: [rng.choice(ext) if rng.random() < 0.85 else rng.randrange(P) for _ in range(k)]
    for log_n in (3, 6, 11, 12, 13):
        c = pick(1 << log_n)
        t = stark.felts_to_tensor(c)
        assert stark.tensor_to_felts(stark.ntt(t)) == S.ntt(c, S.root_of_unity(log_n)), log_n
        assert stark.tensor_to_felts(stark.ntt(t, inverse=True)) == S.intt(c, S.root_of_unity(log_n)), log_n
    for log_n in (4, 9, 11):
        cols = [pick(1 << log_n) for _ in range(2)]
        got = stark.lde(torch.stack([stark.felts_to_tensor(c) for c in cols]))
        for c, g in zip(cols, got):
            assert stark.tensor_to_felts(g) == S.lde(c), log_n
    # a composition is a function of the columns on the coset, whatever they hold: every AIR of the library
    n = 1024
    for air, spec in S.AIRS.items():
        cols = [pick(4 * n) for _ in range(spec["n_cols"])]
        per = stark.periodic_lde(n, air=air)
        exp_per = S.periodic_lde(n, air=air)
        dev_cols = torch.stack([stark.felts_to_tensor(c) for c in cols])
        for trial in range(2):
            alphas = pick(spec["n_constraints"])
            comp = stark.air_eval(dev_cols, per, n, alphas, air=air)
            assert stark.tensor_to_felts(comp) == S.composition_on_coset(cols, exp_per, n, alphas, air=air), (air, trial)
    layer = pick(4 * n)
    t, shift = stark.felts_to_tensor(layer), S.GEN
    while len(layer) > 64:
        beta = rng.choice(ext)
        t, layer = stark.fri_fold(t, beta, shift), S.fri_fold(layer, beta, shift)
        shift = shift * shift % P
        assert stark.tensor_to_felts(t) == layer, len(layer)


def test_commit_rows_matches_oracle(stark):
    import torch
    rng = random.Random(6)
    cols = [[rng.randrange(P) for _ in range(16)] for _ in range(4)]
    t = torch.stack([stark.felts_to_tensor(c) for c in cols])
    assert stark.root_of(stark.commit_rows(t)) == S.commit_rows(cols)
    assert stark.root_of(stark.commit_rows(t[:1])) == S.commit_rows(cols[:1])


def test_full_size_pipeline_properties(stark):
    """2^20-row job: the composition of a valid trace folds down to a low-degree final layer; a
    corrupted trace does not."""
    import torch
    m = 2048
    g = torch.Generator().manual_seed(9)
    xs = torch.randint(0, 2**62, (m, 4), dtype=torch.int64, generator=g)
    ys = torch.randint(0, 2**62, (m, 4), dtype=torch.int64, generator=g)
    xs[:, 3] &= (1 << 58) - 1
    ys[:, 3] &= (1 << 58) - 1
    xs, ys = xs.cuda(), ys.cuda()
    rng = random.Random(10)
    alphas = [rng.randrange(P) for _ in range(S.N_CONSTRAINTS)]
    betas = [rng.randrange(P) for _ in range(16)]
    roots, final = stark.prove_commitments(xs, ys, alphas, betas)
    assert len(roots) == 2 + 15 and len(final) == 64
    shift = S.GEN
    for _ in range(16):
        shift = shift * shift % P
    assert S.poly_degree_bound_check(final, shift, 47)
    # the trace commits to real hashes: row 511 of px is H(x0, y0)
    n = 512 * m
    trace = stark.pedersen_trace(xs, ys)
    x0, y0 = stark.tensor_to_felts(xs[:1])[0], stark.tensor_to_felts(ys[:1])[0]
    assert stark.tensor_to_felts(trace[1][511:512])[0] == R.pedersen_hash(x0, y0)
    # corrupt one cell -> the final layer is no longer low degree
    trace[2][12345][0] += 1
    trace_lde = stark.lde(trace)
    comp = stark.air_eval(trace_lde, stark.periodic_lde(n), n, alphas)
    layer, s = comp, S.GEN
    for k in range(16):
        layer = stark.fri_fold(layer, betas[k], s)
        s = s * s % P
    assert not S.poly_degree_bound_check(stark.tensor_to_felts(layer), s, 47)


def test_configs4_trace_of_2p24_rows_on_one_gpu(stark):
    """BASELINE.json configs[4] at its full size on ONE GPU: 2^24 trace rows (32 768 hashes), LDE to
    2^26 points, both commitments, composition and all 20 folds with their commitments.  Size-independent
    checks: the final layer of the valid trace has degree < 48, the committed trace holds the
    reference's hash of the first input pair, and the trace commitment is the root the level buffer
    ends in (re-derived from two sibling nodes through the scalar hash)."""
    import torch
    m = 1 << 15
    g = torch.Generator().manual_seed(19)
    xs = torch.randint(0, 2**62, (m, 4), dtype=torch.int64, generator=g)
    ys = torch.randint(0, 2**62, (m, 4), dtype=torch.int64, generator=g)
    xs[:, 3] &= (1 << 58) - 1
    ys[:, 3] &= (1 << 58) - 1
    xs, ys = xs.cuda(), ys.cuda()
    rng = random.Random(20)
    alphas = [rng.randrange(P) for _ in range(S.N_CONSTRAINTS)]
    betas = [rng.randrange(P) for _ in range(20)]
    n = 512 * m
    trace = stark.pedersen_trace(xs, ys)
    x0, y0 = stark.tensor_to_felts(xs[:1])[0], stark.tensor_to_felts(ys[:1])[0]
    assert stark.tensor_to_felts(trace[1][511:512])[0] == R.pedersen_hash(x0, y0)
    trace_lde = stark.lde(trace)
    del trace
    lv = stark.commit_rows(trace_lde)
    top = stark.tensor_to_felts(lv[-3:])
    from starkperp import signature
    assert signature.pedersen_hash(top[0], top[1]) == top[2]
    del lv
    comp = stark.air_eval(trace_lde, stark.periodic_lde(n), n, alphas)
    del trace_lde
    roots = [stark.root_of(stark.commit_rows(comp.unsqueeze(0)))]
    layer, s = comp, S.GEN
    for k in range(20):
        layer = stark.fri_fold(layer, betas[k], s)
        s = s * s % P
        if layer.shape[0] > 64:
            roots.append(stark.root_of(stark.commit_rows(layer.unsqueeze(0))))
    assert layer.shape[0] == 64 and len(roots) == 20 and len(set(roots)) == 20
    assert S.poly_degree_bound_check(stark.tensor_to_felts(layer), s, 47)


def test_prove_then_verify_small(stark):
    """GPU prover -> CPU verifier round trip on a 1024-row trace, plus tampering."""
    import copy
    rng = random.Random(21)
    inputs = [(rng.randrange(P), rng.randrange(P)) for _ in range(2)]
    xs = stark.felts_to_tensor([a for a, _ in inputs])
    ys = stark.felts_to_tensor([b for _, b in inputs])
    proof = stark.prove(xs, ys, n_queries=3, seed=7)
    ok, why = S.verify_proof(proof)
    assert ok, why
    bad = copy.deepcopy(proof)
    bad["queries"][0]["trace"][0]["values"][1] ^= 1
    assert S.verify_proof(bad) == (False, "trace path")
    bad = copy.deepcopy(proof)
    bad["queries"][1]["layers"][2][0]["value"] ^= 1
    assert not S.verify_proof(bad)[0]
    bad = copy.deepcopy(proof)
    bad["final_layer"][5] ^= 1
    assert not S.verify_proof(bad)[0]
    # the transcript is chained: changing an EARLIER commitment moves every later challenge, so a proof
    # whose first layer root (or statement) is swapped no longer opens at the drawn query positions
    bad = copy.deepcopy(proof)
    bad["layer_roots"][0] ^= 1
    assert not S.verify_proof(bad)[0]
    bad = copy.deepcopy(proof)
    bad["seed"] += 1
    assert not S.verify_proof(bad)[0]
    bad = copy.deepcopy(proof)
    bad["public_inputs"] = [1]
    assert not S.verify_proof(bad)[0]


def test_prove_then_verify_full_size(stark):
    """2^20-row trace proved on the GPU, verified on the CPU with the C oracle's hash."""
    import torch
    from oracle import cref
    m = 2048
    g = torch.Generator().manual_seed(19)
    xs = torch.randint(0, 2**62, (m, 4), dtype=torch.int64, generator=g)
    ys = torch.randint(0, 2**62, (m, 4), dtype=torch.int64, generator=g)
    xs[:, 3] &= (1 << 58) - 1
    ys[:, 3] &= (1 << 58) - 1
    proof = stark.prove(xs.cuda(), ys.cuda(), n_queries=4, seed=3)
    hash2 = lambda a, b: cref.pedersen_hash_many([a], [b])[0][0]
    ok, why = S.verify_proof(proof, hash2=hash2)
    assert ok, why
    assert len(proof["layer_roots"]) == 16 and len(proof["final_layer"]) == 64


def test_ec_ladder_air_matches_oracle_and_proves(stark):
    """EC-ladder AIR (mimic_ec_mult_air as a trace): GPU witness and composition == oracle;
    the ladder outputs are the reference's mimic_ec_mult_air results; prove -> verify round trip."""
    import torch
    rng = random.Random(33)
    inputs = []
    for _ in range(4):
        q = R.ec_mult(rng.randrange(1, R.EC_ORDER), tuple(R.EC_GEN))
        inputs.append((rng.randrange(1, 2**251), q))
    inputs[1] = (2**251 - 1, tuple(R.EC_GEN))
    inputs[2] = (1, inputs[2][1])
    ms = stark.felts_to_tensor([m for m, _ in inputs])
    qxs = stark.felts_to_tensor([q[0] for _, q in inputs])
    qys = stark.felts_to_tensor([q[1] for _, q in inputs])
    trace = stark.ec_ladder_trace(ms, qxs, qys)
    exp = S.ec_ladder_trace(inputs)
    for g, e in zip(trace, exp):
        assert stark.tensor_to_felts(g) == e
    for k, (m, q) in enumerate(inputs):
        assert (exp[1][256 * k + 251], exp[2][256 * k + 251]) == R.mimic_ec_mult_air(m, q, R.SHIFT_POINT)
    n = 1024
    per = stark.periodic_lde(n, air="ec_ladder")
    exp_per = S.periodic_lde(n, air="ec_ladder")
    for g, e in zip(per, exp_per):
        assert stark.tensor_to_felts(g) == e
    alphas = [rng.randrange(P) for _ in range(S.N_EC_LADDER_CONSTRAINTS)]
    trace_lde = stark.lde(trace)
    comp = stark.air_eval(trace_lde, per, n, alphas, air="ec_ladder")
    exp_comp = S.composition_on_coset([S.lde(c) for c in exp], exp_per, n, alphas, air="ec_ladder")
    assert stark.tensor_to_felts(comp) == exp_comp
    assert S.poly_degree_bound_check(exp_comp, S.GEN, 3 * n - 1)
    proof = stark.prove_ec_ladders(ms, qxs, qys, n_queries=2, seed=5)
    ok, why = S.verify_proof(proof)
    assert ok, why
    proof["queries"][0]["trace"][1]["values"][6] ^= 1
    assert not S.verify_proof(proof)[0]


def test_range_check_air_matches_oracle_and_proves(stark):
    """Range-check AIR: GPU witness, periodic tables and composition == oracle; prove -> verify round trip;
    a value of 2^128 makes the verifier reject (public inputs, and the final-layer degree when they are forged)."""
    rng = random.Random(44)
    values = [0, 1, 2**128 - 1, 2**64] + [rng.randrange(2**128) for _ in range(12)]
    trace = stark.range_check_trace(stark.felts_to_tensor(values))
    exp = S.range_check_trace(values)
    assert stark.tensor_to_felts(trace[0]) == exp[0]
    n = 128 * len(values)
    per = stark.periodic_lde(n, air="range_check")
    exp_per = S.periodic_lde(n, air="range_check")
    for g, e in zip(per, exp_per):
        assert stark.tensor_to_felts(g) == e
    alphas = [rng.randrange(P) for _ in range(S.N_RANGE_CHECK_CONSTRAINTS)]
    comp = stark.air_eval(stark.lde(trace), per, n, alphas, air="range_check")
    exp_comp = S.composition_on_coset([S.lde(c) for c in exp], exp_per, n, alphas, air="range_check")
    assert stark.tensor_to_felts(comp) == exp_comp
    proof = stark.prove_range_checks(values, n_queries=3, seed=9)
    ok, why = S.verify_proof(proof)
    assert ok, why
    bad = stark.prove_range_checks([2**128] + values[1:], n_queries=3, seed=9)
    assert S.verify_proof(bad) == (False, "public inputs")
    bad["public_inputs"][0] = 0  # a prover lying about the statement: the trace itself is not low-degree
    ok, why = S.verify_proof(bad)
    assert not ok
    # 2^10 values (2^17 rows) through the same path
    many = [rng.randrange(2**128) for _ in range(1 << 10)]
    ok, why = S.verify_proof(stark.prove_range_checks(many, n_queries=2, seed=1))
    assert ok, why


def test_block_cyclic_kernels_reassemble_the_single_gpu_columns(stark):
    """The block-cyclic shard kernels of the multi-GPU job, every rank emulated in this process: the
    composition blocks of sp_air_eval_blocks_dev and the folds of sp_fri_fold_blocks_dev, scattered back to
    their global positions, are the single-GPU composition column and fold; sp_interpolate_dev +
    sp_coset_eval_dev give the four cosets of sp_lde_dev; block roots + top = the single tree's root."""
    import torch
    from starkperp import sharded_prover
    ops = sharded_prover.GpuOps("cuda")
    m_hashes = 8
    g = torch.Generator().manual_seed(41)
    xs = torch.randint(0, 2**62, (m_hashes, 4), dtype=torch.int64, generator=g).cuda()
    ys = torch.randint(0, 2**62, (m_hashes, 4), dtype=torch.int64, generator=g).cuda()
    rng = random.Random(42)
    alphas = [rng.randrange(P) for _ in range(S.N_CONSTRAINTS)]
    beta = rng.randrange(P)
    trace = stark.pedersen_trace(xs, ys)
    n = trace.shape[1]
    big = 4 * n
    t_lde = stark.lde(trace)
    # cosets of one interpolation == the LDE
    coef = ops.interpolate(trace)
    w_big = pow(3, (P - 1) // big, P)
    for c in range(4):
        ev = ops.coset_evals(coef[1], 3 * pow(w_big, c, P) % P)
        assert torch.equal(ev, t_lde[1, c::4])
    per = stark.periodic_lde(n, 3, "cuda")
    comp = stark.air_eval(t_lde, per, n, alphas)
    folded = stark.fri_fold(comp, beta, 3)
    want_root = stark.commit_rows(t_lde)[-1]
    for world, log_block in ((4, 5), (8, 7), (2, 11)):
        B = 1 << log_block
        nb_loc = big // B // world
        got_comp = torch.zeros_like(comp)
        got_fold = torch.zeros_like(folded)
        all_roots = torch.zeros((big // B, 4), dtype=torch.int64, device="cuda")
        for rank in range(world):
            blocks = t_lde.view(4, nb_loc, world, B, 4)[:, :, rank]                                  # [4, nb, B, 4]
            halos = torch.roll(t_lde.view(4, big // B, B, 4)[:, :, :4], -1, 1).view(4, nb_loc, world, 4, 4)[:, :, rank]
            shard = torch.cat([blocks, halos], dim=2).contiguous()                                    # [4, nb, B + 4, 4]
            part = ops.air_eval_blocks(shard, per, n.bit_length() - 1, log_block, world, rank, alphas, 3)
            got_comp.view(nb_loc, world, B, 4)[:, rank] = part.view(nb_loc, B, 4)
            all_roots.view(nb_loc, world, 4)[:, rank] = ops.block_roots(blocks.reshape(4, nb_loc * B, 4), log_block)
            if big // 2 >= world * B:
                loc = comp.view(nb_loc, world, B, 4)[:, rank].reshape(nb_loc * B, 4).contiguous()
                h = loc.shape[0] // 2
                f = ops.fold_blocks(loc[:h], loc[h:], big.bit_length() - 1, log_block, world, rank, beta, 3)
                got_fold.view(nb_loc // 2, world, B, 4)[:, rank] = f.view(nb_loc // 2, B, 4)
        assert torch.equal(got_comp, comp), (world, log_block)
        if big // 2 >= world * B:
            assert torch.equal(got_fold, folded), (world, log_block)
        assert torch.equal(ops.merkle_top(all_roots), want_root), (world, log_block)


def test_sharded_prover_on_one_rank_equals_the_plain_job(stark):
    """starkperp.sharded_prover with the library's kernels (GpuOps) and no process group: LDE as 16 coset
    units of 4 interpolations, block-cyclic shard assembly with halos, sp_air_eval_blocks_dev, block roots +
    top - the roots and the final layer must be those of stark.prove_commitments on the same trace (2^14 rows),
    with the default block size and with small blocks."""
    import torch
    from starkperp import sharded_prover
    m = 32
    g = torch.Generator().manual_seed(29)
    xs = torch.randint(0, 2**62, (m, 4), dtype=torch.int64, generator=g)
    ys = torch.randint(0, 2**62, (m, 4), dtype=torch.int64, generator=g)
    xs[:, 3] &= (1 << 58) - 1
    ys[:, 3] &= (1 << 58) - 1
    xs, ys = xs.cuda(), ys.cuda()
    rng = random.Random(30)
    alphas = [rng.randrange(P) for _ in range(S.N_CONSTRAINTS)]
    betas = [rng.randrange(P) for _ in range(10)]
    want_roots, want_final = stark.prove_commitments(xs, ys, alphas, betas)
    roots, final = sharded_prover.commit_job(sharded_prover.GpuOps("cuda"), None, stark.pedersen_trace(xs, ys),
                                             alphas, betas)
    assert roots == want_roots and final == want_final
    roots, final = sharded_prover.commit_job(sharded_prover.GpuOps("cuda"), None, stark.pedersen_trace(xs, ys),
                                             alphas, betas, log_block=6)
    assert roots == want_roots and final == want_final


def _signatures(k, seed):
    from starkperp import batch
    rng = random.Random(seed)
    keys = [rng.randrange(1, R.EC_ORDER) for _ in range(k)]
    zs = [rng.randrange(1, 2**251) for _ in range(k)]
    sigs = batch.sign_many(zs, keys)
    return zs, [r for r, _ in sigs], [s for _, s in sigs], batch.public_keys_many(keys)


def test_ecdsa_air_matches_oracle_and_proves(stark):
    """N4: the ECDSA-verification AIR (three linked ladders).  GPU witness == oracle witness, the third
    ladder's base is zG + rQ, GPU composition == oracle composition, a proof verifies on the CPU and a
    forged public input / tampered opening does not."""
    import copy
    zs, rs, ss, pubs = _signatures(2, 41)
    insts = [S.ecdsa_instance(z, r, s, q) for z, r, s, q in zip(zs, rs, ss, pubs)]
    ws = [i[2] for i in insts]
    trace = stark.ecdsa_trace(*(stark.felts_to_tensor(v) for v in (zs, rs, ws, [q[0] for q in pubs], [q[1] for q in pubs])))
    exp = S.ecdsa_trace(insts)
    for c, (g, e) in enumerate(zip(trace, exp)):
        assert stark.tensor_to_felts(g) == e, "column %d" % c
    n = 2048
    per = stark.periodic_lde(n, air="ecdsa")
    exp_per = S.periodic_lde(n, air="ecdsa")
    for g, e in zip(per, exp_per):
        assert stark.tensor_to_felts(g) == e
    rng = random.Random(42)
    alphas = [rng.randrange(P) for _ in range(S.N_ECDSA_CONSTRAINTS)]
    comp = stark.air_eval(stark.lde(trace), per, n, alphas, air="ecdsa")
    assert stark.tensor_to_felts(comp) == S.composition_on_coset([S.lde(c) for c in exp], exp_per, n, alphas, air="ecdsa")
    proof = stark.prove_ecdsa(zs, rs, ss, pubs, n_queries=2, seed=9)
    ok, why = S.verify_proof(proof)
    assert ok, why
    bad = copy.deepcopy(proof)
    bad["public_inputs"][1] ^= 1          # another r: the transcript (hence every challenge) changes
    assert not S.verify_proof(bad)[0]
    bad = copy.deepcopy(proof)
    bad["queries"][0]["trace"][0]["values"][9] ^= 1
    assert not S.verify_proof(bad)[0]


def test_ecdsa_air_proves_4096_verifications(stark):
    """2^12 signatures = 2^22 trace rows x 10 columns proved on the GPU, verified on the CPU (C-oracle hash)."""
    from oracle import cref
    zs, rs, ss, pubs = _signatures(4096, 43)
    proof = stark.prove_ecdsa(zs, rs, ss, pubs, n_queries=3, seed=2)
    assert proof["n"] == 1 << 22 and len(proof["layer_roots"]) == 18
    ok, why = S.verify_proof(proof, hash2=lambda a, b: cref.pedersen_hash_many([a], [b])[0][0])
    assert ok, why
    # a corrupted signature has no valid witness: the final layer stops being low degree
    ss_bad = list(ss)
    ss_bad[7] = (ss_bad[7] + 1) % R.EC_ORDER
    bad = stark.prove_ecdsa(zs[:8], rs[:8], ss_bad[:8], pubs[:8], n_queries=1, seed=2)
    assert S.verify_proof(bad) == (False, "final layer degree")

"""Test infrastructure: the stage operations of starkperp.sharded_prover on CPU tensors through the
oracle (oracle/stark_ref.py transforms, the C oracle's optimised Pedersen hash for the commitments), so
that the sharding / exchange logic can run under gloo without a GPU."""
import torch

from oracle import cref
from oracle import stark_ref as S

P = S.P


def to_tensor(values):
    import numpy as np
    raw = b"".join(int(v).to_bytes(32, "little") for v in values)
    return torch.from_numpy(np.frombuffer(raw, dtype="<i8").reshape(len(values), 4).copy())


def to_ints(t):
    raw = t.contiguous().numpy().astype("<i8").tobytes()
    return [int.from_bytes(raw[32 * i : 32 * i + 32], "little") for i in range(len(raw) // 32)]


def merkle_root_ints(leaves):
    return cref.opt_merkle_levels(leaves)[-1][0] if len(leaves) > 1 else leaves[0]


def commit_ints(columns):
    leaves = list(columns[0])
    for col in columns[1:]:
        leaves = cref.opt_pedersen_hash_many(leaves, list(col))[0]
    return merkle_root_ints(leaves)


class OracleOps:
    torch = torch

    def empty(self, *shape):
        return torch.zeros(shape, dtype=torch.int64)

    def coset_evals(self, col, shift):
        vals = to_ints(col)
        n = len(vals)
        w = S.root_of_unity(n.bit_length() - 1)
        coeffs = S.intt(vals, w)
        scaled, s = [], 1
        for c in coeffs:
            scaled.append(c * s % P)
            s = s * shift % P
        return to_tensor(S.ntt(scaled, w))

    def commit_root(self, cols):
        return to_tensor([commit_ints([to_ints(c) for c in cols])])[0]

    def merkle_top(self, leaves):
        return to_tensor([merkle_root_ints(to_ints(leaves))])[0]

    def periodic(self, n):
        return S.periodic_lde(n)

    def air_eval_shard(self, shard, per, log_n, row0, alphas, shift):
        n = 1 << log_n
        big = 4 * n
        cols = [to_ints(c) for c in shard]
        m = len(cols[0]) - 4
        w = S.root_of_unity(log_n + 2)
        zinv = [pow((pow(shift, n, P) * pow(w, n * k, P) - 1) % P, -1, P) for k in range(4)]
        out = []
        for i in range(m):
            gi = row0 + i
            cv = S.constraint_values([c[i] for c in cols], [c[i + 4] for c in cols], [t[gi % 2048] for t in per])
            out.append(sum(a * c for a, c in zip(alphas, cv)) % P * zinv[gi % 4] % P)
        assert row0 + m <= big
        return to_tensor(out)

    def fold_shard(self, a, b, log_m, i0, beta, shift):
        av, bv = to_ints(a), to_ints(b)
        w = S.root_of_unity(log_m)
        inv2 = pow(2, -1, P)
        out, x = [], shift * pow(w, i0, P) % P
        for u, v in zip(av, bv):
            out.append(((u + v) * inv2 + beta * (u - v) % P * pow(2 * x, -1, P)) % P)
            x = x * w % P
        return to_tensor(out)

    def to_ints(self, t):
        return to_ints(t)

    def sync(self):
        pass


def single_process_job(inputs, alphas, betas, final_log=6):
    """The reference for the sharded job: the whole pipeline in one process with the oracle."""
    trace = S.pedersen_trace(inputs)
    n = len(trace[0])
    t_lde = [S.lde(col) for col in trace]
    roots = [commit_ints(t_lde)]
    comp = S.composition_on_coset(t_lde, S.periodic_lde(n), n, alphas)
    roots.append(commit_ints([comp]))
    layer, sh, k = comp, S.GEN, 0
    while len(layer) > (1 << final_log):
        layer = S.fri_fold(layer, betas[k], sh)
        sh = sh * sh % P
        k += 1
        if len(layer) > (1 << final_log):
            roots.append(commit_ints([layer]))
    return trace, roots, layer

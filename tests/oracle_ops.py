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

    def interpolate(self, cols):
        out = []
        for col in cols:
            vals = to_ints(col)
            out.append(to_tensor(S.intt(vals, S.root_of_unity(len(vals).bit_length() - 1))))
        return torch.stack(out)

    def coset_evals(self, coef, shift):
        coeffs = to_ints(coef)
        w = S.root_of_unity(len(coeffs).bit_length() - 1)
        scaled, s = [], 1
        for c in coeffs:
            scaled.append(c * s % P)
            s = s * shift % P
        return to_tensor(S.ntt(scaled, w))

    def block_roots(self, cols, log_block):
        columns = [to_ints(c) for c in cols]
        leaves = list(columns[0])
        for col in columns[1:]:
            leaves = cref.opt_pedersen_hash_many(leaves, list(col))[0]
        B = 1 << log_block
        return to_tensor([merkle_root_ints(leaves[i : i + B]) for i in range(0, len(leaves), B)])

    def commit_root(self, cols):
        return to_tensor([commit_ints([to_ints(c) for c in cols])])[0]

    def merkle_top(self, leaves):
        return to_tensor([merkle_root_ints(to_ints(leaves))])[0]

    def periodic(self, n):
        return S.periodic_lde(n)

    def air_eval_blocks(self, shard, per, log_n, log_block, world, rank, alphas, shift):
        n = 1 << log_n
        B = 1 << log_block
        nb = shard.shape[1]
        cols = [to_ints(c.reshape(-1, 4)) for c in shard]  # blocks stored B + 4 rows apart
        w = S.root_of_unity(log_n + 2)
        zinv = [pow((pow(shift, n, P) * pow(w, n * k, P) - 1) % P, -1, P) for k in range(4)]
        out = []
        for t in range(nb):
            for off in range(B):
                ii = t * (B + 4) + off
                gi = (t * world + rank) * B + off
                cv = S.constraint_values([c[ii] for c in cols], [c[ii + 4] for c in cols], [tb[gi % 2048] for tb in per])
                out.append(sum(a * c for a, c in zip(alphas, cv)) % P * zinv[gi % 4] % P)
        return to_tensor(out)

    def fold_blocks(self, a, b, log_m, log_block, world, rank, beta, shift):
        av, bv = to_ints(a), to_ints(b)
        w = S.root_of_unity(log_m)
        inv2 = pow(2, -1, P)
        B = 1 << log_block
        out = []
        for i, (u, v) in enumerate(zip(av, bv)):
            gi = ((i >> log_block) * world + rank) * B + (i & (B - 1))
            x = shift * pow(w, gi, P) % P
            out.append(((u + v) * inv2 + beta * (u - v) % P * pow(2 * x, -1, P)) % P)
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

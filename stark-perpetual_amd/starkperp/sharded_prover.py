"""ONE AIR+FRI commit job over the GPUs of a node (BASELINE.json configs[4]: a
2^24-row trace on 8 MI355X).  One process per GPU; every commitment root equals the root the
single-GPU job (`stark.prove_commitments`) produces for the same trace - the sharding changes where
rows live, never what is hashed.

Stages and their exchange steps (all point-to-point groups = `ncclGroupStart / Send / Recv / End` on
RCCL, plain send / recv on gloo):

  LDE       The blowup-4 coset LDE of a column is four independent size-n coset transforms
            (out[4 j + c] = f(shift * w_4n^c * w_n^j)), so the 4 columns give 16 units that are spread
            over the ranks with no communication ("column-sharded iNTT / LDE").  ONE bulk all-to-all then
            turns unit outputs into LDE-ROW shards: rank r receives rows [r M/N, (r + 1) M/N) of every
            column plus a halo of one trace row (4 LDE rows) for the `next row` reads of the AIR.
  commit    per-shard row chains and subtree, all_gather of N x 32 B sub-roots, log2 N top levels on
            every rank (a Pedersen hash is not an ncclRedOp: "tree-reduce" = all_gather + local top).
  AIR       row-local on the shard + halo (sp_air_eval_shard_dev).
  FRI       the layers stay in natural order (so that every layer root equals the single-GPU one); a
            fold pairs i with i + M/2, i.e. the new shard of rank g needs half a shard from rank g // 2
            and half a shard from rank g // 2 + N / 2: one two-peer exchange per fold (each layer crosses
            the links once).  Once a layer is down to `tail_rows` rows per rank it is all-gathered and the
            rest of the folds and commits run replicated on every rank.

`ops` hides the device: `GpuOps` calls the library (the product path); tests plug in the oracle on CPU
tensors to check the orchestration with gloo.  Build-defined like the rest of the prover: parity unpinned,
every hash is the pinned pedersen_hash.
"""
from typing import List, Sequence

FIELD_PRIME = 2**251 + 17 * 2**192 + 1
FIELD_GEN = 3
BLOWUP = 4


def root_of_unity(log_n: int) -> int:
    return pow(FIELD_GEN, (FIELD_PRIME - 1) >> log_n, FIELD_PRIME)


class GpuOps:
    """The stage kernels through the C ABI (include/starkperp.h); tensors are int64 [rows, 4] in HBM."""

    def __init__(self, device="cuda"):
        import torch
        from . import _lib, stark
        self.torch, self._lib, self.stark, self.device = torch, _lib, stark, device
        self.lib = _lib.ensure_init()

    def empty(self, *shape):
        return self.torch.empty(shape, dtype=self.torch.int64, device=self.device)

    def coset_evals(self, col, shift):
        """[n, 4] evaluations over <w_n> -> [n, 4] evaluations of the interpolant over shift * <w_n>."""
        return self.stark.lde(col.unsqueeze(0), 0, shift)[0]

    def commit_root(self, cols):
        """cols [W, m, 4] -> [4] root of the subtree over the m row leaves."""
        return self.stark.commit_rows(cols)[-1]

    def merkle_top(self, leaves):
        """[w, 4] sub-roots -> [4] root (w a power of two)."""
        w = leaves.shape[0]
        if w == 1:
            return leaves[0]
        buf = self.empty(2 * w - 1, 4)
        buf[:w] = leaves
        self._lib.check(self.lib.sp_merkle_build_dev(buf.data_ptr(), w.bit_length() - 1, None, self._stream()),
                        "sp_merkle_build_dev")
        return buf[-1]

    def periodic(self, n):
        return self.stark.periodic_lde(n, FIELD_GEN, self.device)

    def air_eval_shard(self, shard, per, log_n, row0, alphas, shift):
        """shard [4, m + 4, 4] (m rows + halo) -> [m, 4] composition values at global rows row0 .. row0 + m."""
        m = shard.shape[1] - 4
        out = self.empty(m, 4)
        self._lib.check(self.lib.sp_air_eval_shard_dev(
            shard.data_ptr(), shard.shape[1], m, row0, per.data_ptr(), log_n, self._lib.pack_felts(alphas),
            self._lib.pack_felts([shift]), out.data_ptr(), self._stream()), "sp_air_eval_shard_dev")
        return out

    def fold_shard(self, a, b, log_m, i0, beta, shift):
        cnt = a.shape[0]
        out = self.empty(cnt, 4)
        self._lib.check(self.lib.sp_fri_fold_shard_dev(
            a.data_ptr(), b.data_ptr(), out.data_ptr(), log_m, i0, cnt, self._lib.pack_felts([beta]),
            self._lib.pack_felts([shift]), self._stream()), "sp_fri_fold_shard_dev")
        return out

    def to_ints(self, t):
        return self.stark.tensor_to_felts(t)

    def _stream(self):
        return self.torch.cuda.current_stream().cuda_stream

    def sync(self):
        self.torch.cuda.current_stream().synchronize()


def _staged(dist, t):
    """gloo moves host memory only: device tensors are staged through the host there (CPU tests and the
    one-GPU two-rank test); RCCL takes the device pointers as they are."""
    return t.is_cuda and dist.get_backend() == "gloo"


def _exchange(dist, torch, sends, recvs, rank):
    """sends: [(dst, tensor)], recvs: [(src, tensor)] in an order both sides agree on; messages to self are
    copied.  One grouped batch of point-to-point operations."""
    local_out = [t for dst, t in sends if dst == rank]
    local_in = [t for src, t in recvs if src == rank]
    assert len(local_out) == len(local_in)
    for o, i in zip(local_out, local_in):
        i.copy_(o)
    if dist is None:
        return
    out_t = [(dst, t.cpu() if _staged(dist, t) else t) for dst, t in sends if dst != rank]
    in_t = [(src, t, torch.empty(t.shape, dtype=t.dtype) if _staged(dist, t) else t) for src, t in recvs if src != rank]
    ops = [dist.P2POp(dist.isend, t, dst) for dst, t in out_t]
    ops += [dist.P2POp(dist.irecv, buf, src) for src, _, buf in in_t]
    if ops:
        for req in dist.batch_isend_irecv(ops):
            req.wait()
    for _, t, buf in in_t:
        if buf is not t:
            t.copy_(buf)


def _all_gather_rows(dist, torch, rows, world):
    """[k, 4] per rank -> [world * k, 4] in rank order."""
    if dist is None or world == 1:
        return rows
    src = rows.contiguous()
    if _staged(dist, src):
        host = src.cpu()
        parts = [torch.empty_like(host) for _ in range(world)]
        dist.all_gather(parts, host)
        return torch.cat(parts).to(rows.device)
    parts = [torch.empty_like(src) for _ in range(world)]
    dist.all_gather(parts, src)
    return torch.cat(parts)


def commit_job(ops, dist, trace_cols, alphas: Sequence[int], betas: Sequence[int], shift: int = FIELD_GEN,
               final_log: int = 6, tail_rows: int = 1024):
    """trace_cols: [W, n, 4] full trace columns (every rank holds the columns of the units it owns; simplest
    is all of them).  Returns (roots, final_layer) as Python ints on every rank:
    roots = [trace, composition, fri_1, ...] exactly like stark.prove_commitments."""
    torch = ops.torch
    P = FIELD_PRIME
    world = dist.get_world_size() if dist is not None else 1
    rank = dist.get_rank() if dist is not None else 0
    assert world & (world - 1) == 0 and world <= 16
    ncols, n = trace_cols.shape[0], trace_cols.shape[1]
    log_n = n.bit_length() - 1
    big = BLOWUP * n
    m = big // world      # LDE rows per rank
    jn = n // world       # trace-domain positions per rank
    assert jn >= 512, "a rank's shard must cover whole periods of the periodic columns"
    w_big = root_of_unity(log_n + 2)

    # ---- LDE units -> row shards ------------------------------------------------------------------
    units = [(col, c) for col in range(ncols) for c in range(BLOWUP)]
    sends, recvs, keep = [], [], []
    for u, (col, c) in enumerate(units):
        if u % world == rank:
            ev = ops.coset_evals(trace_cols[col], shift * pow(w_big, c, P) % P)  # [n, 4]: LDE rows 4 j + c
            for r in range(world):
                j0 = r * jn
                halo = (j0 + jn) % n
                chunk = torch.cat([ev[j0 : j0 + jn], ev[halo : halo + 1]])
                keep.append(chunk)
                sends.append((r, chunk))
    shard = ops.empty(ncols, m + 4, 4)
    bufs = []
    for u, (col, c) in enumerate(units):
        buf = ops.empty(jn + 1, 4)
        bufs.append(buf)
        recvs.append((u % world, buf))
    if dist is not None and world > 1:
        ops.sync()
    _exchange(dist if world > 1 else None, torch, sends, recvs, rank)
    view = shard.view(ncols, jn + 1, BLOWUP, 4)
    for (col, c), buf in zip(units, bufs):
        view[col, :, c, :] = buf
    del keep, bufs, sends, recvs

    def combine(local_root):
        subs = _all_gather_rows(dist, torch, local_root.reshape(1, 4), world)
        return ops.merkle_top(subs)

    roots = [combine(ops.commit_root(shard[:, :m].contiguous()))]
    # ---- composition on the shard -------------------------------------------------------------------
    per = ops.periodic(n)
    comp = ops.air_eval_shard(shard, per, log_n, rank * m, list(alphas), shift)
    del shard
    roots.append(combine(ops.commit_root(comp.unsqueeze(0))))

    # ---- FRI: sharded folds, then the replicated tail ---------------------------------------------
    layer, size, sh, k = comp, big, shift, 0
    sharded = world > 1
    while size > (1 << final_log):
        if sharded and (size // 2) // world < tail_rows:
            if dist is not None:
                ops.sync()
            layer = _all_gather_rows(dist, torch, layer, world)
            sharded = False
        if sharded:
            mk = size // world          # rows per rank in this layer
            half = mk // 2              # rows per rank in the next one
            # old rank q: its half h goes to new rank 2 (q mod N/2) + h, as `a` from the lower half of the
            # layer (q < N/2) or as `b` from the upper half
            sends = [(2 * (rank % (world // 2)) + h, layer[h * half : (h + 1) * half].contiguous()) for h in (0, 1)]
            a, b = ops.empty(half, 4), ops.empty(half, 4)
            recvs = [(rank // 2, a), (rank // 2 + world // 2, b)]
            ops.sync()
            # both sides order their messages by the sender's rank, then by half
            _exchange(dist, torch, sends, recvs, rank)
            layer = ops.fold_shard(a, b, size.bit_length() - 1, rank * half, betas[k], sh)
        else:
            full_half = size // 2
            layer = ops.fold_shard(layer[:full_half], layer[full_half:], size.bit_length() - 1, 0, betas[k], sh)
        size //= 2
        sh = sh * sh % P
        k += 1
        if size > (1 << final_log):
            local = ops.commit_root(layer.unsqueeze(0))
            roots.append(combine(local) if sharded else local)
    final = ops.to_ints(layer)
    return [ops.to_ints(r.reshape(1, 4))[0] for r in roots], final

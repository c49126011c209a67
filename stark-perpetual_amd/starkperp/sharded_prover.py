"""ONE AIR+FRI commit job over the GPUs of a node (BASELINE.json configs[4]: a
2^24-row trace on 8 MI355X).  One process per GPU; every commitment root equals the root the
single-GPU job (`stark.prove_commitments`) produces for the same trace - the sharding changes where
rows live, never what is hashed.

Row layout: BLOCK-CYCLIC.  The LDE rows (and later the positions of every FRI layer) are cut into blocks
of B = 2^log_block consecutive indices and block b belongs to rank b mod N; local block t of rank r is
global block t N + r.  Three things follow:

  AIR       the `next row` read (i + blowup) stays inside a block except for its last trace row: every
            block carries a halo of 4 LDE rows, shipped with the block in the one bulk exchange below.
  FRI       a fold pairs i with i + M/2, and M / (2 B) is a multiple of N while M >= 2 B N: BOTH members of
            every pair are on the same rank ("the first log2(M/8) folds are shard-local"; the
            block-cyclic natural order is what the bit-reversed contiguous order is there) and the folded
            layer is block-cyclic again - NO data moves between folds.  When a layer is down to one block
            per rank (B N points) it is all-gathered once and the remaining folds run replicated.
  commits   stay in natural order, so every root is the single-GPU root: the bottom log_block levels of a
            tree lie inside blocks (per-rank forest of block subtrees), the M / B block roots are
            all-gathered (32 B each: 1 MiB for the first layer of a 2^24-row trace at B = 2^11) and every
            rank hashes the top of the tree (a Pedersen hash is not an ncclRedOp: "tree-reduce" =
            all_gather + local top).

Exchange steps (point-to-point groups = `ncclGroupStart / Send / Recv / End` on RCCL, plain send / recv
on gloo):

  LDE       The blowup-4 coset LDE of a column is four independent size-n coset transforms
            (out[4 j + c] = f(shift * w_4n^c * w_n^j)) of ONE interpolation, so the 4 columns give 16
            units; consecutive units go to the same rank (8 ranks: the two cosets a rank owns share their
            column's interpolation).  ONE bulk all-to-all turns unit outputs into the block-cyclic row
            shards (with the halos): 4 x 4n x 32 B in all, (N - 1) / N of it over the links.
  roots     one all_gather of block roots per commitment.
  tail      one all_gather of B N felts.

`commit_job` takes the constraint and folding challenges as inputs: it is the COMMIT phase of the
benchmark job (BASELINE.json's metric), not a prover - `stark.prove*` derives its challenges from a
transcript of the roots; threading that through here means drawing alpha after roots[0] and beta_k after
roots[k + 1].

`ops` hides the device: `GpuOps` calls the library (the product path); tests plug in the oracle on CPU
tensors to check the orchestration with gloo.  Build-defined like the rest of the prover: parity unpinned,
every hash is the pinned pedersen_hash.
"""
from typing import Sequence

FIELD_PRIME = 2**251 + 17 * 2**192 + 1
FIELD_GEN = 3
BLOWUP = 4
DEFAULT_LOG_BLOCK = 11


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

    def interpolate(self, cols):
        """[k, n, 4] evaluations over <w_n> -> [k, n, 4] coefficients (bit-reversed order: only
        `coset_evals` reads them)."""
        cols = cols.contiguous()
        k, n = cols.shape[0], cols.shape[1]
        coef = self.empty(k, n, 4)
        self._lib.check(self.lib.sp_interpolate_dev(cols.data_ptr(), coef.data_ptr(), k, n.bit_length() - 1,
                                                    self._stream()), "sp_interpolate_dev")
        return coef

    def coset_evals(self, coef, shift):
        """[n, 4] coefficients of `interpolate` -> [n, 4] evaluations over shift * <w_n>."""
        n = coef.shape[0]
        out = self.empty(n, 4)
        self._lib.check(self.lib.sp_coset_eval_dev(coef.data_ptr(), out.data_ptr(), 1, n.bit_length() - 1,
                                                   self._lib.pack_felts([shift]), self._stream()), "sp_coset_eval_dev")
        return out

    def block_roots(self, cols, log_block):
        """cols [W, m, 4] -> [m >> log_block, 4]: roots of the subtrees over blocks of 2^log_block row leaves
        (leaf = left-fold chain of the row's felts; one column commits the felts themselves)."""
        cols = cols.contiguous()
        w, m = cols.shape[0], cols.shape[1]
        nb = m >> log_block
        levels = self.empty(nb * ((2 << log_block) - 1), 4)
        if w == 1:
            levels[:m] = cols[0]
        else:
            self._lib.check(self.lib.sp_pedersen_chains_dev(cols.data_ptr(), m, w, levels.data_ptr(), None,
                                                            self._stream()), "sp_pedersen_chains_dev")
        if log_block > 0:
            self._lib.check(self.lib.sp_merkle_forest_dev(levels.data_ptr(), nb, log_block, None, self._stream()),
                            "sp_merkle_forest_dev")
        return levels[levels.shape[0] - nb:]

    def commit_root(self, cols):
        """cols [W, m, 4] -> [4] root of the tree over the m row leaves."""
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

    def air_eval_blocks(self, shard, per, log_n, log_block, world, rank, alphas, shift):
        """shard [4, nb, B + 4, 4] (blocks with their halos) -> [nb * B, 4] composition values."""
        nb = shard.shape[1]
        out = self.empty(nb << log_block, 4)
        self._lib.check(self.lib.sp_air_eval_blocks_dev(
            shard.data_ptr(), shard.shape[1] * shard.shape[2], nb, log_block, world, rank, per.data_ptr(), log_n,
            self._lib.pack_felts(alphas), self._lib.pack_felts([shift]), out.data_ptr(), self._stream()),
            "sp_air_eval_blocks_dev")
        return out

    def fold_blocks(self, a, b, log_m, log_block, world, rank, beta, shift):
        cnt = a.shape[0]
        out = self.empty(cnt, 4)
        self._lib.check(self.lib.sp_fri_fold_blocks_dev(
            a.data_ptr(), b.data_ptr(), out.data_ptr(), log_m, cnt, log_block, world, rank,
            self._lib.pack_felts([beta]), self._lib.pack_felts([shift]), self._stream()), "sp_fri_fold_blocks_dev")
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
    _p2p_batch(dist, torch, [(dst, t) for dst, t in sends if dst != rank], [(src, t) for src, t in recvs if src != rank])


def _p2p_batch(dist, torch, out_msgs, in_msgs):
    """One grouped batch of point-to-point operations (ncclGroupStart / Send / Recv / End under RCCL): out_msgs
    [(dst, tensor)], in_msgs [(src, tensor)]; device tensors go as they are except under gloo (staged)."""
    out_t = [(dst, t.cpu() if _staged(dist, t) else t) for dst, t in out_msgs]
    in_t = [(src, t, torch.empty(t.shape, dtype=t.dtype) if _staged(dist, t) else t) for src, t in in_msgs]
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
    if dist is None:
        return rows
    src = rows.contiguous()  # (a world of one still goes through the backend: bench.py --force-dist rehearses RCCL so)
    if _staged(dist, src):
        host = src.cpu()
        parts = [torch.empty_like(host) for _ in range(world)]
        dist.all_gather(parts, host)
        return torch.cat(parts).to(rows.device)
    parts = [torch.empty_like(src) for _ in range(world)]
    dist.all_gather(parts, src)
    return torch.cat(parts)


def unit_owner(u: int, n_units: int, world: int) -> int:
    """Consecutive units (the cosets of one column) go to the same rank; with more ranks than units the
    last ranks own none."""
    per = -(-n_units // world)
    return u // per


def commit_job(ops, dist, trace_cols, alphas: Sequence[int], betas: Sequence[int], shift: int = FIELD_GEN,
               final_log: int = 6, log_block: int = None, stats: dict = None):
    """trace_cols: [W, n, 4] full trace columns (a rank reads only the columns of the units it owns).
    Returns (roots, final_layer) as Python ints on every rank: roots = [trace, composition, fri_1, ...]
    exactly like stark.prove_commitments.  `stats`, when given, receives the bytes this rank sent."""
    torch = ops.torch
    P = FIELD_PRIME
    world = dist.get_world_size() if dist is not None else 1
    rank = dist.get_rank() if dist is not None else 0
    assert world & (world - 1) == 0, "the job needs a power-of-two number of ranks"
    ncols, n = trace_cols.shape[0], trace_cols.shape[1]
    log_n = n.bit_length() - 1
    big = BLOWUP * n
    if log_block is None:
        log_block = DEFAULT_LOG_BLOCK
    log_block = max(2, min(log_block, (big // world).bit_length() - 1))
    B = 1 << log_block
    B4 = B // BLOWUP          # trace-domain positions per block
    nb_tot = big // B         # blocks in all
    assert nb_tot % world == 0 and nb_tot >= world, "every rank needs at least one block of LDE rows"
    nb_loc = nb_tot // world  # blocks per rank
    m = nb_loc * B            # LDE rows per rank
    w_big = root_of_unity(log_n + 2)
    sent = 0

    # ---- LDE units -> block-cyclic row shards (with halos): the one bulk exchange ---------------------
    units = [(col, c) for col in range(ncols) for c in range(BLOWUP)]
    mine = [u for u in range(len(units)) if unit_owner(u, len(units), world) == rank]
    my_cols = sorted({units[u][0] for u in mine})
    coef = ops.interpolate(torch.stack([trace_cols[c] for c in my_cols])) if my_cols else None
    sends, recvs, keep = [], [], []
    for u in mine:
        col, c = units[u]
        ev = ops.coset_evals(coef[my_cols.index(col)], shift * pow(w_big, c, P) % P)  # [n, 4]: LDE rows 4 j + c
        blocks = ev.view(nb_loc, world, B4, 4)
        # halo of block b = the first trace position of block b + 1 (cyclically): the `next row` of its last one
        halos = torch.roll(ev.view(nb_tot, B4, 4)[:, 0], -1, 0).view(nb_loc, world, 4)
        for r in range(world):
            chunk = torch.cat([blocks[:, r], halos[:, r].unsqueeze(1)], dim=1).contiguous()  # [nb_loc, B4 + 1, 4]
            keep.append(chunk)
            sends.append((r, chunk))
            if r != rank:
                sent += chunk.numel() * 8
    bufs = []
    for u in range(len(units)):
        buf = ops.empty(nb_loc, B4 + 1, 4)
        bufs.append(buf)
        recvs.append((unit_owner(u, len(units), world), buf))
    if dist is not None and world > 1:
        ops.sync()
    _exchange(dist if world > 1 else None, torch, sends, recvs, rank)
    shard = ops.empty(ncols, nb_loc, B + BLOWUP, 4)
    view = shard.view(ncols, nb_loc, B4 + 1, BLOWUP, 4)
    for (col, c), buf in zip(units, bufs):
        view[col, :, :, c, :] = buf  # one strided copy per unit
    del keep, bufs, sends, recvs, coef

    def combine(local_roots):
        """[nb, 4] block roots of this rank (local block t = global block t N + r) -> the root of the layer."""
        nb = local_roots.shape[0]
        if dist is not None and world > 1:
            ops.sync()
        allr = _all_gather_rows(dist, torch, local_roots, world)          # [world * nb, 4], rank-major
        subs = allr.view(world, nb, 4).transpose(0, 1).reshape(world * nb, 4) if world > 1 else allr
        return ops.merkle_top(subs.contiguous())

    rows = shard[:, :, :B].reshape(ncols, m, 4)
    roots = [combine(ops.block_roots(rows, log_block))]
    sent += (world - 1) * nb_loc * 32 if world > 1 else 0
    del rows
    # ---- composition on the shard -------------------------------------------------------------------
    per = ops.periodic(n)
    comp = ops.air_eval_blocks(shard, per, log_n, log_block, world, rank, list(alphas), shift)
    del shard
    roots.append(combine(ops.block_roots(comp.unsqueeze(0), log_block)))
    sent += (world - 1) * nb_loc * 32 if world > 1 else 0

    # ---- FRI: shard-local folds, then the replicated tail ---------------------------------------------
    layer, size, sh, k = comp, big, shift, 0
    sharded = world > 1
    while size > (1 << final_log):
        if sharded and size // 2 < world * B:
            # one block per rank is left: gathered in rank order it is the layer in natural order
            ops.sync()
            sent += (world - 1) * layer.shape[0] * 32
            layer = _all_gather_rows(dist, torch, layer, world)
            sharded = False
        half = layer.shape[0] // 2
        if sharded:
            layer = ops.fold_blocks(layer[:half], layer[half:], size.bit_length() - 1, log_block, world, rank,
                                    betas[k], sh)
        else:
            layer = ops.fold_shard(layer[:half], layer[half:], size.bit_length() - 1, 0, betas[k], sh)
        size //= 2
        sh = sh * sh % P
        k += 1
        if size > (1 << final_log):
            if sharded:
                roots.append(combine(ops.block_roots(layer.unsqueeze(0), log_block)))
                sent += (world - 1) * (layer.shape[0] >> log_block) * 32
            else:
                roots.append(ops.commit_root(layer.unsqueeze(0)))
    if sharded:
        # A final layer that is still sharded (a caller-chosen final_log with 2^final_log >= world * B: the loop
        # ended before the one-block-per-rank gather; ADVICE r3 - round 3 returned this rank's slice as "the" final
        # layer).  Every rank holds blocks r, r + N, r + 2 N, ...: gather and put them back in natural order.
        ops.sync()
        nb = layer.shape[0] >> log_block
        sent += (world - 1) * layer.shape[0] * 32
        allr = _all_gather_rows(dist, torch, layer, world)                  # [world * nb * B, 4], rank-major
        layer = allr.view(world, nb, B, 4).transpose(0, 1).reshape(world * nb * B, 4).contiguous()
        sharded = False
    final = ops.to_ints(layer)
    if stats is not None:
        stats.update({"bytes_sent_by_this_rank": sent, "log_block": log_block, "blocks_per_rank": nb_loc,
                      "units_owned": len(mine), "interpolations": len(my_cols)})
    return [ops.to_ints(r.reshape(1, 4))[0] for r in roots], final

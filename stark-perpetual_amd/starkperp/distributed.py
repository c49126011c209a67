"""One process per GPU.  The hot path shards by construction (independent hashes / signatures /
subtrees - SURVEY.md section 8(e)); the only exchange step is the combine of per-rank Merkle
sub-roots: an all_gather of world_size x 32 bytes (RCCL over xGMI when the backend is "nccl",
gloo in the CPU tests) followed by the log2(world_size) top levels hashed redundantly on every
rank.  A Pedersen hash is not an ncclRedOp, so the "tree-reduce" is all_gather + local top-of-tree.
"""
from typing import Callable, List, Sequence


def shard_range(n_units: int, rank: int, world: int):
    """Contiguous [lo, hi) slice of n_units owned by `rank` (remainder spread over low ranks)."""
    base, rem = divmod(n_units, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def felt_to_tensor(torch, value: int, device=None):
    limbs = [(value >> (64 * i)) & ((1 << 64) - 1) for i in range(4)]
    signed = [l - (1 << 64) if l >= (1 << 63) else l for l in limbs]
    return torch.tensor(signed, dtype=torch.int64, device=device)


def tensor_to_felt(t) -> int:
    vals = [int(v) & ((1 << 64) - 1) for v in t.tolist()]
    return sum(v << (64 * i) for i, v in enumerate(vals))


def gather_subroots(dist, torch, subroot: int, device=None, group=None) -> List[int]:
    """all_gather of one felt per rank; returns the sub-roots in rank order."""
    world = dist.get_world_size(group)
    mine = felt_to_tensor(torch, subroot, device)
    out = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(out, mine, group=group)
    return [tensor_to_felt(t) for t in out]


def combine_subroots(subroots: Sequence[int],
                     hash_many: Callable[[Sequence[int], Sequence[int]], List[int]]) -> int:
    """Top of the tree over world_size (a power of two) sub-roots."""
    level = list(subroots)
    assert len(level) >= 1 and len(level) & (len(level) - 1) == 0
    while len(level) > 1:
        level = hash_many(level[0::2], level[1::2])
    return level[0]


def sharded_merkle_root(dist, torch, local_leaves: Sequence[int], merkle_root, hash_many,
                        device=None, group=None) -> int:
    """Root of the tree whose leaves are the concatenation of every rank's `local_leaves`
    (equal power-of-two counts): local subtree -> all_gather -> top levels."""
    sub = merkle_root(local_leaves)
    return combine_subroots(gather_subroots(dist, torch, sub, device, group), hash_many)


def combine_subroots_dev(lib, dist, subroot_row, gathered, top, stream):
    """Device-resident variant used by bench.py: `subroot_row` is the [4] int64 view of this
    rank's root in HBM, `gathered` a [world, 4] buffer, `top` a [2*world-1, 4] buffer."""
    from . import _lib
    world = gathered.shape[0]
    dist.all_gather_into_tensor(gathered, subroot_row.reshape(1, 4))
    top[:world].copy_(gathered)
    height = world.bit_length() - 1
    assert 1 << height == world
    _lib.check(lib.sp_merkle_build_dev(top.data_ptr(), height, None, stream), "sp_merkle_build_dev")
    return top[-1]


# ---- forests: nb independent trees per rank, combined tree by tree ------------------------------
def forest_gather_layout(gathered_rank_major, world: int, nb: int):
    """`gathered_rank_major[r * nb + t]` = sub-root of tree t on rank r (what all_gather returns)
    -> tree-major leaves `[t * world + r]` of the nb top trees (contiguous leaves per tree)."""
    return [gathered_rank_major[r * nb + t] for t in range(nb) for r in range(world)]


def combine_forest_subroots(dist, torch, local_roots: Sequence[int], hash_many, device=None, group=None):
    """Host-int version: every rank holds the sub-roots of nb trees; returns the nb job roots
    (tree t = the sub-roots of tree t from every rank, in rank order)."""
    world = dist.get_world_size(group)
    nb = len(local_roots)
    mine = torch.stack([felt_to_tensor(torch, v, device) for v in local_roots])
    out = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(out, mine, group=group)
    rank_major = [tensor_to_felt(row) for t in out for row in t]
    leaves = forest_gather_layout(rank_major, world, nb)
    return [combine_subroots(leaves[t * world : (t + 1) * world], hash_many) for t in range(nb)]


def combine_forest_dev(lib, dist, roots, gathered, top, nb: int, stream):
    """Device-resident version used by bench.py.  roots: [nb, 4] view of this rank's sub-roots;
    gathered: >= [world * nb, 4]; top: >= [nb * (2 * world - 1), 4].  One all_gather (RCCL) of
    world * nb * 32 bytes, then the nb top trees (any nb) as one lockstep forest of height log2(world)."""
    from . import _lib
    world = dist.get_world_size()
    g = gathered[: world * nb]
    dist.all_gather_into_tensor(g, roots)
    top[: world * nb] = g.reshape(world, nb, 4).transpose(0, 1).reshape(world * nb, 4)
    _lib.check(lib.sp_merkle_forest_dev(top.data_ptr(), nb, world.bit_length() - 1, None, stream),
               "sp_merkle_forest_dev")
    return top[nb * (2 * world - 1) - nb : nb * (2 * world - 1)]


# ---- multi-update sharded by key prefix ("Merkle multi-update (C3)") ----------------
class ShardedSparseTree:
    """A height-h sparse tree over `world` ranks: rank g owns the subtree of the keys whose top
    log2(world) bits equal g (a tree of height h - log2(world) over the remaining low bits).  An
    update is local work on the owned subtree plus the usual exchange: an all_gather of one
    sub-root per rank and the log2(world) top levels hashed on every rank.

    `make_tree(height, empty_leaf)` builds the local tree - `state.LibrarySparseTree` on the GPU,
    `state.SparseMerkleTree` with an injected hash in the CPU tests; `hash_many` hashes the top."""

    def __init__(self, dist, torch, height: int, make_tree, hash_many, empty_leaf: int = 0, device=None,
                 group=None):
        self.dist, self.torch, self.device, self.group = dist, torch, device, group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.top_bits = self.world.bit_length() - 1
        assert 1 << self.top_bits == self.world and self.top_bits < height
        self.height = height
        self.low_bits = height - self.top_bits
        self.local = make_tree(self.low_bits, empty_leaf)
        self.hash_many = hash_many
        self.root = self._combine(self.local.root)

    def owner(self, key: int) -> int:
        return key >> self.low_bits

    def _combine(self, local_root: int) -> int:
        return combine_subroots(gather_subroots(self.dist, self.torch, local_root, self.device, self.group),
                                self.hash_many)

    def update(self, modifications) -> tuple:
        """Every rank passes the whole batch {key: leaf} (or at least its own share); each applies
        the keys it owns.  Returns (old_root, new_root) of the full tree on every rank."""
        mask = (1 << self.low_bits) - 1
        mine = {k & mask: v for k, v in dict(modifications).items() if self.owner(k) == self.rank}
        old_root = self.root
        _, local_new = self.local.update(mine)
        self.root = self._combine(local_new)
        return old_root, self.root

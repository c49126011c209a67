"""Leaf / tree conventions of the perpetual state (the callers either side of the hash kernels).

  * position_hash: services/perpetual/cairo/position/hash.cairo:22-74 (bounds
    definitions/constants.cairo:11-38);
  * orders-tree leaf = fulfilled amount felt, order id = top 64 bits of the 251-bit message hash
    (services/perpetual/cairo/order/order.cairo:23-59,122-124);
  * tree updates: merkle_multi_update call sites state/state.cairo:155-173.
All hashing goes through starkperp.batch (GPU)."""
from typing import Dict, Iterable, List, Sequence, Tuple

from . import batch

BALANCE_LOWER_BOUND = -(2**63)
BALANCE_UPPER_BOUND = 2**63
FUNDING_INDEX_LOWER_BOUND = -(2**63)
FUNDING_INDEX_UPPER_BOUND = 2**63
N_ASSETS_UPPER_BOUND = 2**16

Position = Tuple[int, int, Sequence[Tuple[int, int, int]]]  # (public_key, collateral, assets)


def pack_asset(asset_id: int, cached_funding_index: int, balance: int) -> int:
    """hash.cairo:30-36."""
    packed = asset_id
    packed = packed * (FUNDING_INDEX_UPPER_BOUND - FUNDING_INDEX_LOWER_BOUND) + (
        cached_funding_index - FUNDING_INDEX_LOWER_BOUND)
    return packed * (BALANCE_UPPER_BOUND - BALANCE_LOWER_BOUND) + (balance - BALANCE_LOWER_BOUND)


def position_words(position: Position) -> List[int]:
    """The chain  H(...H(H(0, a_1), a_2)..., public_key), tail)  as its input words."""
    public_key, collateral, assets = position
    tail = (collateral - BALANCE_LOWER_BOUND) * N_ASSETS_UPPER_BOUND + len(assets)
    return [0] + [pack_asset(*a) for a in assets] + [public_key, tail]


def position_hash(position: Position) -> int:
    """hash.cairo:58-74."""
    return batch.pedersen_chain(position_words(position))


def position_hashes_many(positions: Iterable[Position]) -> List[int]:
    """Many position leaves: chains are grouped by depth (n_assets + 3 words) and each group runs
    as batched launches."""
    words = [position_words(p) for p in positions]
    out = [0] * len(words)
    by_depth: Dict[int, List[int]] = {}
    for i, w in enumerate(words):
        by_depth.setdefault(len(w), []).append(i)
    for depth, idxs in by_depth.items():
        res = batch.pedersen_chains_many([words[i] for i in idxs])
        for i, r in zip(idxs, res):
            out[i] = r
    return out


def order_id_of(message_hash: int) -> int:
    """order/order.cairo:23-59: the 64 most significant bits of the 251-bit hash."""
    return message_hash >> 187


def orders_tree_root(fulfilled: Dict[int, int], height: int = 64) -> int:
    """Root of the orders tree (leaf = fulfilled amount, empty leaf 0) after writing `fulfilled`."""
    return batch.merkle_sparse_root(height, fulfilled, 0)


# ---- persistent sparse tree with a preimage ("facts") store -------------------------------------
class SparseMerkleTree:
    """Height-h Pedersen Merkle tree over 2^h leaves, stored sparsely as the reference stores it:
    a map node_hash -> (left, right) (`merkle_facts`, services/perpetual/cairo/main.cairo:39-40,
    61-64) plus the current root.  `update` is the equivalent of one
    merkle_multi_update{hash_ptr=pedersen_ptr} call (state/state.cairo:155-173): it walks the
    subtree induced by the modified leaves (starkware/python/merkle_tree.py:4-26), takes untouched
    siblings from the store and recomputes the touched nodes level by level - one batched GPU
    launch pair per level through `hash_many` (default: starkperp.batch.pedersen_hash_many).

    Only bookkeeping (dicts of ints) happens on the host; every hash goes through `hash_many`.
    """

    def __init__(self, height: int, empty_leaf: int = 0, hash_many=None):
        self.height = height
        self.hash_many = hash_many or batch.pedersen_hash_many
        self.facts: Dict[int, Tuple[int, int]] = {}
        self.empties = [empty_leaf]
        for _ in range(height):
            prev = self.empties[-1]
            node = self.hash_many([prev], [prev])[0]
            self.facts[node] = (prev, prev)
            self.empties.append(node)
        self.root = self.empties[height]

    def _children(self, node: int, level: int) -> Tuple[int, int]:
        """Children of `node`, which sits `level` levels above the leaves."""
        if node == self.empties[level]:
            e = self.empties[level - 1]
            return e, e
        return self.facts[node]

    def get(self, key: int) -> int:
        node = self.root
        for level in range(self.height, 0, -1):
            left, right = self._children(node, level)
            node = right if (key >> (level - 1)) & 1 else left
        return node

    def get_many(self, keys: Sequence[int]) -> List[int]:
        return [self.get(k) for k in keys]

    def update(self, modifications: Dict[int, int]) -> Tuple[int, int]:
        """Writes {leaf_index: value}; returns (old_root, new_root)."""
        old_root = self.root
        if not modifications:
            return old_root, old_root
        for k in modifications:
            assert 0 <= k < (1 << self.height)
        # top-down: current hashes of every node on a modified path, per level (level = height
        # above the leaves), keyed by node index within its level
        paths: List[Dict[int, int]] = [dict() for _ in range(self.height + 1)]
        paths[self.height][0] = self.root
        for level in range(self.height, 0, -1):
            wanted = sorted(set(k >> (level - 1) for k in modifications))
            for idx in wanted:
                parent = paths[level][idx >> 1]
                left, right = self._children(parent, level)
                paths[level - 1][idx] = right if idx & 1 else left
                sib = idx ^ 1
                if sib not in paths[level - 1]:
                    paths[level - 1][sib] = left if idx & 1 else right
        # bottom-up: new values
        layer = dict(modifications)
        for level in range(1, self.height + 1):
            parents = sorted(set(i >> 1 for i in layer))
            lefts = [layer.get(2 * i, paths[level - 1][2 * i]) for i in parents]
            rights = [layer.get(2 * i + 1, paths[level - 1][2 * i + 1]) for i in parents]
            hashes = self.hash_many(lefts, rights)
            for node, l, r in zip(hashes, lefts, rights):
                self.facts[node] = (l, r)
            layer = dict(zip(parents, hashes))
        self.root = layer[0]
        return old_root, self.root


class LibrarySparseTree:
    """The same tree with its state kept by the library (sp_tree_*, csrc/merkle.hip): one call per
    update instead of one host round trip per level - 4096 leaves at height 64 in about 15 ms
    instead of 270.  Same interface as SparseMerkleTree (`update`, `get`, `root`); node preimages
    are not exposed (the library stores nodes by position, not by hash)."""

    def __init__(self, height: int, empty_leaf: int = 0, context: int = 0):
        """context: which device of sp_init_devices keeps the tree (0 = the primary; one process per GPU has only that)."""
        import ctypes
        from . import _lib
        self._lib, self._ct = _lib, ctypes
        self.height = height
        handle = ctypes.c_int()
        _lib.check(_lib.ensure_init().sp_tree_create_on(context, height, _lib.pack_felts([empty_leaf]),
                                                        ctypes.byref(handle)), "sp_tree_create_on")
        self._handle = handle.value

    @property
    def root(self) -> int:
        out = self._lib.new_felts(1)
        self._lib.check(self._lib.ensure_init().sp_tree_root(self._handle, out), "sp_tree_root")
        return self._lib.unpack_felts(out, 1)[0]

    def get(self, key: int) -> int:
        out = self._lib.new_felts(1)
        keys = (self._ct.c_uint64 * 1)(key)
        self._lib.check(self._lib.ensure_init().sp_tree_get(self._handle, keys, 1, out), "sp_tree_get")
        return self._lib.unpack_felts(out, 1)[0]

    def get_many(self, keys: Sequence[int]) -> List[int]:
        n = len(keys)
        if n == 0:
            return []
        out = self._lib.new_felts(n)
        arr = (self._ct.c_uint64 * n)(*keys)
        self._lib.check(self._lib.ensure_init().sp_tree_get(self._handle, arr, n, out), "sp_tree_get")
        return self._lib.unpack_felts(out, n)

    def update(self, modifications: Dict[int, int]) -> Tuple[int, int]:
        """Writes {leaf_index: value}; returns (old_root, new_root)."""
        items = sorted(dict(modifications).items())
        n = len(items)
        for k, v in items:
            assert 0 <= k < (1 << self.height) and 0 <= v < batch.FIELD_PRIME
        keys = (self._ct.c_uint64 * max(n, 1))(*[k for k, _ in items])
        old, new, st = self._lib.new_felts(1), self._lib.new_felts(1), self._lib.new_bytes(1)
        self._lib.check(self._lib.ensure_init().sp_tree_update(
            self._handle, keys, self._lib.pack_felts([v for _, v in items]), n, old, new, st), "sp_tree_update")
        if st[0]:
            raise AssertionError("Unhashable input." if st[0] & 2 else "leaf out of range")
        return self._lib.unpack_felts(old, 1)[0], self._lib.unpack_felts(new, 1)[0]

    def update_arrays(self, keys, leaves) -> Tuple[int, int]:
        """The same update from NumPy arrays: keys uint64[n] (any order, distinct), leaves uint64[n, 4]."""
        import numpy as np
        keys = np.ascontiguousarray(keys, dtype=np.uint64)
        leaves = np.ascontiguousarray(leaves, dtype=np.uint64)
        assert keys.ndim == 1 and leaves.shape == (keys.shape[0], 4)
        order = np.argsort(keys, kind="stable")
        keys, leaves = np.ascontiguousarray(keys[order]), np.ascontiguousarray(leaves[order])
        old, new, st = self._lib.new_felts(1), self._lib.new_felts(1), self._lib.new_bytes(1)
        self._lib.check(self._lib.ensure_init().sp_tree_update(
            self._handle, keys.ctypes.data_as(self._ct.c_void_p), leaves.ctypes.data_as(self._ct.c_void_p),
            keys.shape[0], old, new, st), "sp_tree_update")
        if st[0]:
            raise AssertionError("Unhashable input." if st[0] & 2 else "leaf out of range")
        return self._lib.unpack_felts(old, 1)[0], self._lib.unpack_felts(new, 1)[0]

    def close(self):
        if self._handle is not None:
            handle, self._handle = self._handle, None
            lib = self._lib.load()
            if lib.sp_is_initialised():  # after sp_shutdown the library has already dropped every tree
                self._lib.check(lib.sp_tree_destroy(handle), "sp_tree_destroy")

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001 - interpreter teardown / library already gone
            pass


def hash_position_updates(updates: Sequence[Tuple[int, Position, Position]]):
    """position/hash.cairo:76-131: (key, prev_position, new_position) -> (key, prev_hash, new_hash);
    an unchanged position is hashed once."""
    prev = position_hashes_many([u[1] for u in updates])
    changed = [i for i, u in enumerate(updates) if u[1] != u[2]]
    new_h = position_hashes_many([updates[i][2] for i in changed])
    out = [(u[0], p, p) for u, p in zip(updates, prev)]
    for i, hnew in zip(changed, new_h):
        out[i] = (updates[i][0], prev[i], hnew)
    return out


# ---- the per-batch state-root update ------------------------------------------------------------
def squash_updates(accesses: Sequence[Tuple[int, object, object]]):
    """squash_dict semantics (state/state.cairo:67-96): a chronological list of
    (key, prev_value, new_value) accesses -> one (key, first_prev, last_new) per key, sorted by key;
    every access must continue from the previous value of its key."""
    first, last = {}, {}
    for key, prev, new in accesses:
        if key in last:
            assert last[key] == prev, "inconsistent dict access for key %d" % key
        else:
            first[key] = prev
        last[key] = new
    return [(k, first[k], last[k]) for k in sorted(first)]


class SharedState:
    """The two roots of services/perpetual/cairo/state/state.cairo:99-107 with their sparse trees
    (leaf of the positions tree = position_hash, empty leaf = hash of the empty position; leaf of
    the orders tree = fulfilled amount, empty leaf 0)."""

    EMPTY_POSITION: Position = (0, 0, ())

    def __init__(self, positions_tree_height: int = 64, orders_tree_height: int = 64, hash_many=None,
                 position_hashes=None):
        self._position_hashes = position_hashes or position_hashes_many
        empty_leaf = self._position_hashes([self.EMPTY_POSITION])[0]
        if hash_many is None:  # the library keeps the trees: one call per update
            self.positions = LibrarySparseTree(positions_tree_height, empty_leaf)
            self.orders = LibrarySparseTree(orders_tree_height, 0)
        else:  # an injected hash (the oracle's, in CPU tests): host bookkeeping, hashes through it
            self.positions = SparseMerkleTree(positions_tree_height, empty_leaf, hash_many)
            self.orders = SparseMerkleTree(orders_tree_height, 0, hash_many)

    @property
    def positions_root(self) -> int:
        return self.positions.root

    @property
    def orders_root(self) -> int:
        return self.orders.root

    def apply_state_updates(self, position_accesses, order_accesses):
        """shared_state_apply_state_updates (state/state.cairo:135-186): squash, hash the previous
        and new positions (hash_position_updates), check the previous leaves against the tree,
        merkle-multi-update both trees.  Returns ((old_pos_root, new_pos_root), (old_ord, new_ord))."""
        # Every precondition first - the Cairo twin fails the whole batch, so nothing may be written
        # before both access lists have been squashed and checked against both trees.
        pos = squash_updates(position_accesses)
        orders = squash_updates(order_accesses)
        prev_h = self._position_hashes([p for _, p, _ in pos])
        changed = [i for i, (_, p, q) in enumerate(pos) if p != q]
        new_h = list(prev_h)
        for i, hv in zip(changed, self._position_hashes([pos[i][2] for i in changed])):
            new_h[i] = hv
        assert self.positions.get_many([key for key, _, _ in pos]) == list(prev_h), \
            "previous position does not match the tree"
        assert self.orders.get_many([key for key, _, _ in orders]) == [prev for _, prev, _ in orders], \
            "previous order state does not match the tree"
        for key, _, new in orders:
            assert 0 <= key < (1 << self.orders.height) and 0 <= new < batch.FIELD_PRIME, \
                "order leaf out of range"
        pos_roots = self.positions.update({k: hv for (k, _, _), hv in zip(pos, new_h)})
        try:
            ord_roots = self.orders.update({k: new for k, _, new in orders})
        except BaseException:
            # data-dependent failure inside the second update (an unhashable node): put the previous
            # position leaves back so that the two roots still describe one batch boundary
            self.positions.update({k: hv for (k, _, _), hv in zip(pos, prev_h)})
            raise
        return pos_roots, ord_roots

    def close(self):
        """Releases the library-side trees (no-op for the host-bookkeeping variant)."""
        for tree in (self.positions, self.orders):
            if hasattr(tree, "close"):
                tree.close()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

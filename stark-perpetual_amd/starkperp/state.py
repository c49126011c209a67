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

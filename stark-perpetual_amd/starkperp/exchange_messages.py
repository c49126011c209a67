"""Multi-asset order message: the one signed-message format of the exchange that
has no Python twin in the reference - its only statement is the Cairo function
services/exchange/cairo/signature_message_hashes.cairo:387-471 (helpers :171-329).

Pinning status: signature_test_data.json holds a `multi_asset_order` fixture (:102-139) with a
`message_hash` (:185-188).  The (message_hash, private_key, r, s) part of it is reproduced (sign and
verify, tests/golden/g8_reference_fixtures_extra.json), but the hash itself does NOT follow from the
fixture's fields under the Cairo source of this tree - nor under 18 000 layout variants that were
searched (list order, index base, count order, packing group, initial value) - so the fixture was made
by another revision of the format.  The packing below is therefore restated from the Cairo source
alone: parity unpinned for the packing, every hash inside it is the pinned pedersen_hash."""
from typing import Callable, Sequence, Tuple

from . import batch
from .signature import pedersen_hash

MULTI_ASSET_OFFCHAIN_ORDER_TYPE = 6  # signature_message_hashes.cairo:21
LIST_FIELD_BOUND = 2**12  # MULTI_ASSET_ORDER_LIST_FIELD_SIZE_UPPER_BOUND / N_CONDITIONS_UPPER_BOUND, :22-23
INDICES_PER_FELT = 20  # :218-222
VaultInfo = Tuple[int, int, int, int]  # (vault_id, public_key, asset_id, amount), field order of :172-178


def multi_asset_order_words(signer_key: int, nonce: int, expiration_timestamp: int, system_id: int,
                            give: Sequence[VaultInfo], receive: Sequence[VaultInfo],
                            conditions: Sequence[int]):
    """Felts of the hash chain in chain order: conditions, assets (receive then give), third-party
    keys, (vault, amount) fields three per felt, third-party indices twenty per felt, packed metadata."""
    assert 0 <= nonce < 2**32 and 0 <= expiration_timestamp < 2**32 and 0 <= system_id < 2**126
    assert len(give) < LIST_FIELD_BOUND and len(receive) < LIST_FIELD_BOUND and len(conditions) < LIST_FIELD_BOUND
    fields, assets, third_keys, third_idx = [], [], [], []
    for entries in (receive, give):  # :405-418
        for index, (vault_id, public_key, asset_id, amount) in enumerate(entries):
            assert 0 <= vault_id < 2**64 and 0 <= amount < 2**64
            assets.append(asset_id)
            fields += [vault_id, amount]
            if public_key != signer_key:  # :302-309
                third_idx.append(index)
                third_keys.append(public_key)
    words = list(conditions) + assets + third_keys
    for i in range(0, len(fields), 3):  # :264-288
        acc = 0
        for v in fields[i : i + 3]:
            acc = (acc << 64) + v
        words.append(acc)
    for i in range(0, len(third_idx), INDICES_PER_FELT):  # :205-260
        acc = 0
        for v in third_idx[i : i + INDICES_PER_FELT]:
            acc = (acc << 12) + v
        words.append(acc)
    meta = MULTI_ASSET_OFFCHAIN_ORDER_TYPE  # :433-463
    for value, width in ((nonce, 32), (expiration_timestamp, 32), (len(give), 12), (len(receive), 12),
                         (len(third_idx), 12), (len(conditions), 12), (system_id, 126)):
        meta = (meta << width) + value
    words.append(meta << 3)
    return words


def multi_asset_order_hash(signer_key: int, nonce: int, expiration_timestamp: int, system_id: int,
                           give: Sequence[VaultInfo], receive: Sequence[VaultInfo], conditions: Sequence[int],
                           hash_function: Callable[..., int] = pedersen_hash) -> int:
    """signature_message_hashes.cairo:387-471: left fold of the words from words[0]
    (hash_felts_no_padding with initial_hash = the first felt, :465-470)."""
    words = multi_asset_order_words(signer_key, nonce, expiration_timestamp, system_id, give, receive,
                                    conditions)
    acc = words[0]
    for w in words[1:]:
        acc = hash_function(acc, w)
    return acc


def multi_asset_order_msgs_many(orders: Sequence[Sequence]):
    """Many multi-asset orders (7-tuples in multi_asset_order_hash argument order).  Chains have
    different lengths, so orders are grouped by chain length and every group is hashed as one
    depth-d, width-n batch on the GPU."""
    words = [multi_asset_order_words(*o) for o in orders]
    out = [None] * len(orders)
    by_len = {}
    for i, w in enumerate(words):
        by_len.setdefault(len(w), []).append(i)
    for length, idx in by_len.items():
        if length == 1:
            for i in idx:
                out[i] = words[i][0]
            continue
        for i, h in zip(idx, batch.pedersen_chains_many([words[i] for i in idx])):
            out[i] = h
    return out

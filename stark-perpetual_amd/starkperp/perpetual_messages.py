"""Signed-message builders with the names, argument order and bounds of the reference's
services/perpetual/public/perpetual_messages.py (Cairo twins:
services/exchange/cairo/signature_message_hashes.cairo:56-170).  The scalar functions take the same
`hash_function=` injection seam; `*_many` variants hash whole order batches on the GPU as
depth-d, width-n chains (sp_pedersen_chains_dev shape)."""
from typing import Callable, Sequence

from . import batch
from .keccak import keccak256
from .signature import pedersen_hash

LIMIT_ORDER_WITH_FEES = 3
TRANSFER = 4
CONDITIONAL_TRANSFER = 5
WITHDRAWAL = 6
WITHDRAWAL_TO_ADDRESS = 7


def build_condition(fact_registry_address: str, fact: bytes) -> int:
    """Condition felt of a conditional transfer (perpetual_messages.py:15-21): Keccak-256 over the
    packed (address, bytes32) pair, cut to its 250 low bits.  web3's solidityKeccak is replaced by
    the host Keccak in starkperp.keccak; the address is 20 bytes of hex with or without "0x"."""
    digits = fact_registry_address[2:] if fact_registry_address[:2] in ("0x", "0X") else fact_registry_address
    address = bytes.fromhex(digits)
    if len(address) != 20:
        raise ValueError("fact registry address must be 20 bytes, got %d" % len(address))
    if len(fact) != 32:
        raise ValueError("fact must be 32 bytes, got %d" % len(fact))
    return int.from_bytes(keccak256(address + bytes(fact)), "big") & (2**250 - 1)


def _limit_order_words(asset_id_synthetic, asset_id_collateral, is_buying_synthetic, asset_id_fee,
                       amount_synthetic, amount_collateral, max_amount_fee, nonce, position_id,
                       expiration_timestamp):
    if is_buying_synthetic:
        sell, buy, n_sell, n_buy = asset_id_collateral, asset_id_synthetic, amount_collateral, amount_synthetic
    else:
        sell, buy, n_sell, n_buy = asset_id_synthetic, asset_id_collateral, amount_synthetic, amount_collateral
    word0 = ((n_sell * 2**64 + n_buy) * 2**64 + max_amount_fee) * 2**32 + nonce
    word1 = LIMIT_ORDER_WITH_FEES
    for _ in range(3):
        word1 = word1 * 2**64 + position_id
    word1 = (word1 * 2**32 + expiration_timestamp) * 2**17
    return [sell, buy, asset_id_fee, word0, word1]


def _transfer_words(kind, sender_position_id, receiver_position_id, src_fee_position_id, nonce,
                    amount, max_amount_fee, expiration_timestamp):
    word0 = ((sender_position_id * 2**64 + receiver_position_id) * 2**64 + src_fee_position_id) * 2**32 + nonce
    word1 = (((kind * 2**64 + amount) * 2**64 + max_amount_fee) * 2**32 + expiration_timestamp) * 2**81
    return word0, word1


def _withdrawal_word(position_id, nonce, amount, expiration_timestamp, kind=WITHDRAWAL_TO_ADDRESS):
    word = kind
    word = word * 2**64 + position_id
    word = word * 2**32 + nonce
    word = word * 2**64 + amount
    word = word * 2**32 + expiration_timestamp
    return word * 2**49


def _fold(hash_function, words):
    acc = hash_function(words[0], words[1])
    for w in words[2:]:
        acc = hash_function(acc, w)
    return acc


def get_limit_order_msg_without_bounds(asset_id_synthetic, asset_id_collateral, is_buying_synthetic,
                                       asset_id_fee, amount_synthetic, amount_collateral,
                                       max_amount_fee, nonce, position_id, expiration_timestamp,
                                       hash_function: Callable[..., int] = pedersen_hash) -> int:
    """perpetual_messages.py:253-286."""
    return _fold(hash_function, _limit_order_words(
        asset_id_synthetic, asset_id_collateral, is_buying_synthetic, asset_id_fee, amount_synthetic,
        amount_collateral, max_amount_fee, nonce, position_id, expiration_timestamp))


def get_limit_order_msg(asset_id_synthetic, asset_id_collateral, is_buying_synthetic, asset_id_fee,
                        amount_synthetic, amount_collateral, max_amount_fee, nonce, position_id,
                        expiration_timestamp, hash_function: Callable[..., int] = pedersen_hash) -> int:
    """perpetual_messages.py:212-250."""
    assert 0 <= asset_id_synthetic < 2**128
    assert 0 <= asset_id_collateral < 2**250
    assert 0 <= asset_id_fee < 2**250
    assert 0 <= amount_synthetic < 2**64
    assert 0 <= amount_collateral < 2**64
    assert 0 <= max_amount_fee < 2**64
    assert 0 <= nonce < 2**32
    assert 0 <= position_id < 2**64
    assert 0 <= expiration_timestamp < 2**32
    return get_limit_order_msg_without_bounds(
        asset_id_synthetic, asset_id_collateral, is_buying_synthetic, asset_id_fee, amount_synthetic,
        amount_collateral, max_amount_fee, nonce, position_id, expiration_timestamp,
        hash_function=hash_function)


def get_transfer_msg_without_bounds(asset_id, asset_id_fee, receiver_public_key, sender_position_id,
                                    receiver_position_id, src_fee_position_id, nonce, amount,
                                    max_amount_fee, expiration_timestamp,
                                    hash_function: Callable[..., int] = pedersen_hash) -> int:
    """perpetual_messages.py:136-162."""
    w0, w1 = _transfer_words(TRANSFER, sender_position_id, receiver_position_id, src_fee_position_id,
                             nonce, amount, max_amount_fee, expiration_timestamp)
    return _fold(hash_function, [asset_id, asset_id_fee, receiver_public_key, w0, w1])


def get_transfer_msg(asset_id, asset_id_fee, receiver_public_key, sender_position_id,
                     receiver_position_id, src_fee_position_id, nonce, amount, max_amount_fee,
                     expiration_timestamp, hash_function: Callable[..., int] = pedersen_hash) -> int:
    """perpetual_messages.py:97-133."""
    assert 0 <= amount < 2**64
    assert 0 <= asset_id < 2**250
    assert 0 <= asset_id_fee < 2**250
    assert 0 <= expiration_timestamp < 2**32
    assert 0 <= max_amount_fee < 2**64
    assert 0 <= nonce < 2**32
    assert 0 <= receiver_position_id < 2**64
    assert 0 <= receiver_public_key < 2**251
    assert 0 <= sender_position_id < 2**64
    assert 0 <= src_fee_position_id < 2**64
    return get_transfer_msg_without_bounds(
        asset_id, asset_id_fee, receiver_public_key, sender_position_id, receiver_position_id,
        src_fee_position_id, nonce, amount, max_amount_fee, expiration_timestamp,
        hash_function=hash_function)


def get_conditional_transfer_msg_without_bounds(asset_id, asset_id_fee, receiver_public_key, condition,
                                                sender_position_id, receiver_position_id,
                                                src_fee_position_id, nonce, amount, max_amount_fee,
                                                expiration_timestamp,
                                                hash_function: Callable[..., int] = pedersen_hash) -> int:
    """perpetual_messages.py:66-94."""
    w0, w1 = _transfer_words(CONDITIONAL_TRANSFER, sender_position_id, receiver_position_id,
                             src_fee_position_id, nonce, amount, max_amount_fee, expiration_timestamp)
    return _fold(hash_function, [asset_id, asset_id_fee, receiver_public_key, condition, w0, w1])


def get_conditional_transfer_msg(asset_id, asset_id_fee, receiver_public_key, condition,
                                 sender_position_id, receiver_position_id, src_fee_position_id, nonce,
                                 amount, max_amount_fee, expiration_timestamp,
                                 hash_function: Callable[..., int] = pedersen_hash) -> int:
    """perpetual_messages.py:24-63."""
    assert 0 <= amount < 2**64
    assert 0 <= asset_id < 2**250
    assert 0 <= asset_id_fee < 2**250
    assert 0 <= condition < 2**251
    assert 0 <= expiration_timestamp < 2**32
    assert 0 <= src_fee_position_id < 2**64
    assert 0 <= max_amount_fee < 2**64
    assert 0 <= nonce < 2**32
    assert 0 <= receiver_position_id < 2**64
    assert 0 <= receiver_public_key < 2**251
    assert 0 <= sender_position_id < 2**64
    return get_conditional_transfer_msg_without_bounds(
        asset_id, asset_id_fee, receiver_public_key, condition, sender_position_id,
        receiver_position_id, src_fee_position_id, nonce, amount, max_amount_fee,
        expiration_timestamp, hash_function=hash_function)


def get_withdrawal_to_address_msg_without_bounds(asset_id_collateral, position_id, eth_address, nonce,
                                                 expiration_timestamp, amount,
                                                 hash_function: Callable[..., int] = pedersen_hash) -> int:
    """perpetual_messages.py:192-209."""
    return _fold(hash_function, [asset_id_collateral, int(eth_address, 16),
                                 _withdrawal_word(position_id, nonce, amount, expiration_timestamp)])


def get_withdrawal_to_address_msg(asset_id_collateral, position_id, eth_address, nonce,
                                  expiration_timestamp, amount,
                                  hash_function: Callable[..., int] = pedersen_hash) -> int:
    """perpetual_messages.py:165-189."""
    assert 0 <= asset_id_collateral < 2**250
    assert 0 <= nonce < 2**32
    assert 0 <= position_id < 2**64
    assert 0 <= expiration_timestamp < 2**32
    assert 0 <= amount < 2**64
    assert 0 <= int(eth_address, 16) < 2**160
    return get_withdrawal_to_address_msg_without_bounds(
        asset_id_collateral, position_id, eth_address, nonce, expiration_timestamp, amount,
        hash_function=hash_function)


def get_withdrawal_msg_without_bounds(asset_id_collateral, position_id, nonce, expiration_timestamp, amount,
                                      hash_function: Callable[..., int] = pedersen_hash) -> int:
    """The old-API withdrawal (transaction type 6: the owner key IS the signing key and is not part of
    the message): services/perpetual/cairo/transactions/withdrawal.cairo:57-60,66-74, JS twin
    services/perpetual/public/js/perpetual_messages.js:49-82.  h(asset_id_collateral,
    6 | position_id (64) | nonce (32) | amount (64) | expiration_timestamp (32) | 0 (49))."""
    return hash_function(asset_id_collateral,
                         _withdrawal_word(position_id, nonce, amount, expiration_timestamp, WITHDRAWAL))


def get_withdrawal_msg(asset_id_collateral, position_id, nonce, expiration_timestamp, amount,
                       hash_function: Callable[..., int] = pedersen_hash) -> int:
    """Argument order of the JS builder (perpetual_messages.js:49-56); bounds :63-72 (the Cairo
    assumptions withdrawal.cairo:42-46)."""
    assert 0 <= asset_id_collateral < 2**250
    assert 0 <= nonce < 2**32
    assert 0 <= position_id < 2**64
    assert 0 <= expiration_timestamp < 2**32
    assert 0 <= amount < 2**64
    return get_withdrawal_msg_without_bounds(
        asset_id_collateral, position_id, nonce, expiration_timestamp, amount, hash_function=hash_function)


def withdrawal_hash(asset_id_collateral, position_id, owner_key, public_key, nonce, expiration_timestamp,
                    amount, hash_function: Callable[..., int] = pedersen_hash) -> int:
    """The message the Cairo program checks a withdrawal's signature against (withdrawal.cairo:47-78):
    type 6 without the owner key when owner_key == public_key (the old API), otherwise type 7 over
    h(asset_id_collateral, owner_key) - the owner key in the place of the eth address."""
    if owner_key == public_key:
        return get_withdrawal_msg_without_bounds(
            asset_id_collateral, position_id, nonce, expiration_timestamp, amount, hash_function=hash_function)
    return _fold(hash_function, [asset_id_collateral, owner_key,
                                 _withdrawal_word(position_id, nonce, amount, expiration_timestamp)])


def get_price_msg(oracle_name: int, asset_pair: int, timestamp: int, price: int,
                  hash_function=pedersen_hash):
    """perpetual_messages.py:311-326."""
    assert 0 <= oracle_name < 2**40
    assert 0 <= asset_pair < 2**128
    assert 0 <= timestamp < 2**32
    assert 0 <= price < 2**120
    return hash_function((asset_pair << 40) + oracle_name, (price << 32) + timestamp)


# ---- batch additions --------------------------------------------------------------------------
def limit_order_msgs_many(orders: Sequence[Sequence[int]]):
    """Message hashes of many limit orders (each a 10-tuple in get_limit_order_msg argument
    order): four GPU launches of width len(orders) instead of 4 * len(orders) scalar hashes."""
    words = [_limit_order_words(*o) for o in orders]
    return batch.pedersen_chains_many(words)


def transfer_msgs_many(transfers: Sequence[Sequence[int]]):
    """Many transfers (10-tuples in get_transfer_msg argument order): depth-5 chains."""
    words = []
    for (asset_id, asset_id_fee, receiver_public_key, sender, receiver, fee_pos, nonce, amount,
         max_fee, expiration) in transfers:
        w0, w1 = _transfer_words(TRANSFER, sender, receiver, fee_pos, nonce, amount, max_fee, expiration)
        words.append([asset_id, asset_id_fee, receiver_public_key, w0, w1])
    return batch.pedersen_chains_many(words)


def conditional_transfer_msgs_many(transfers: Sequence[Sequence[int]]):
    """Many conditional transfers (11-tuples in get_conditional_transfer_msg order): depth 6."""
    words = []
    for (asset_id, asset_id_fee, receiver_public_key, condition, sender, receiver, fee_pos, nonce,
         amount, max_fee, expiration) in transfers:
        w0, w1 = _transfer_words(CONDITIONAL_TRANSFER, sender, receiver, fee_pos, nonce, amount, max_fee,
                                 expiration)
        words.append([asset_id, asset_id_fee, receiver_public_key, condition, w0, w1])
    return batch.pedersen_chains_many(words)


def price_msgs_many(prices: Sequence[Sequence[int]]):
    """Many oracle price messages ((oracle_name, asset_pair, timestamp, price) tuples): the
    1-hash + 1-verify per oracle signature of oracle/oracle_price.cairo:96-108."""
    xs = [(asset_pair << 40) + oracle for oracle, asset_pair, _, _ in prices]
    ys = [(price << 32) + ts for _, _, ts, price in prices]
    return batch.pedersen_hash_many(xs, ys)


def withdrawal_to_address_msgs_many(withdrawals: Sequence[Sequence]):
    """Many withdrawals to an address (6-tuples in get_withdrawal_to_address_msg argument order,
    eth_address a hex string or an int): depth-3 chains."""
    words = []
    for asset_id_collateral, position_id, eth_address, nonce, expiration, amount in withdrawals:
        address = int(eth_address, 16) if isinstance(eth_address, str) else int(eth_address)
        words.append([asset_id_collateral, address, _withdrawal_word(position_id, nonce, amount, expiration)])
    return batch.pedersen_chains_many(words)


def withdrawal_msgs_many(withdrawals: Sequence[Sequence[int]]):
    """Many old-API withdrawals (5-tuples in get_withdrawal_msg argument order): one hash each."""
    xs = [w[0] for w in withdrawals]
    ys = [_withdrawal_word(position_id, nonce, amount, expiration, WITHDRAWAL)
          for _, position_id, nonce, expiration, amount in withdrawals]
    return batch.pedersen_hash_many(xs, ys)


def withdrawal_hashes_many(withdrawals: Sequence[Sequence[int]]):
    """withdrawal_hash for a mixed batch (7-tuples: asset_id_collateral, position_id, owner_key,
    public_key, nonce, expiration_timestamp, amount): the type-6 messages as one batch of single hashes,
    the type-7 ones as depth-3 chains; results in input order."""
    old = [i for i, w in enumerate(withdrawals) if w[2] == w[3]]
    new = [i for i, w in enumerate(withdrawals) if w[2] != w[3]]
    out = [None] * len(withdrawals)
    if old:
        got = withdrawal_msgs_many([(withdrawals[i][0], withdrawals[i][1], withdrawals[i][4],
                                     withdrawals[i][5], withdrawals[i][6]) for i in old])
        for i, v in zip(old, got):
            out[i] = v
    if new:
        got = batch.pedersen_chains_many([
            [withdrawals[i][0], withdrawals[i][2],
             _withdrawal_word(withdrawals[i][1], withdrawals[i][4], withdrawals[i][6], withdrawals[i][5])]
            for i in new])
        for i, v in zip(new, got):
            out[i] = v
    return out


def verify_price_signatures_many(prices: Sequence[Sequence[int]], signatures: Sequence[Sequence[int]],
                                 signer_keys: Sequence[int]):
    """Oracle price quorum check (oracle/oracle_price.cairo:96-108): one message hash and one
    signature verification per signed price, both batched.  prices[i] = (oracle_name, asset_pair,
    timestamp, price), signatures[i] = (r, s), signer_keys[i] = the oracle's x-only Stark key.
    Returns one bool per signed price (False for anything the reference's verify would reject or
    assert on)."""
    if not (len(prices) == len(signatures) == len(signer_keys)):
        raise ValueError("prices, signatures and signer_keys must have the same length")
    messages = price_msgs_many(prices)
    codes = batch.verify_codes(messages, [r for r, _ in signatures], [s for _, s in signatures],
                               list(signer_keys))
    return [c == 1 for c in codes]

"""Serial Pedersen hash-chain consumers.  Chains cannot be
parallelised - each link needs the previous canonical x - so these run as one single-hash GPU
launch per link (~0.1 ms each); they are here so that the harnesses built on the hash keep working
on the same backend.

  * config hashes: services/perpetual/public/generate_perpetual_config_hash.py:73-175 (Cairo twin
    services/perpetual/cairo/definitions/general_config_hash.cairo:46-148): left fold from 0 over
    the field values followed by their count;
  * program hash: starkware/cairo/bootloaders/program_hash_test_utils.py:7-21 calls cairo-lang's
    compute_program_hash_chain, which is not in the reference tree; the shape below restates public
    cairo-lang behaviour (hash chain over [len, bootloader_version, main, n_builtins, builtins...,
    data...]) and is PARITY UNPINNED here (no compiled program, no cairo-lang to compare with).
"""
from typing import Iterable, List, Sequence, Union

from . import batch

HASH_BYTES = 32
IntLike = Union[int, bool, str]


def convert2int(val: IntLike) -> int:
    """generate_perpetual_config_hash.py:43-53: decimal string, hex string, bool or int."""
    if type(val) in (int, bool):
        return int(val)
    assert type(val) is str, "Unsupported type."
    if len(val) > 2 and val[:2] == "0x":
        return int(val, 16)
    return int(val, 10)


def hash_chain_from_zero(values: Iterable[IntLike]) -> int:
    """h = 0; for v: h = pedersen_hash(h, v)   (generate_perpetual_config_hash.py:118-121)."""
    return batch.pedersen_chain([0] + [convert2int(v) for v in values])


def general_config_hash(config: dict, hash_version: IntLike) -> bytes:
    """calculate_general_config_hash (:73-122).  `hash_version` is GENERAL_CONFIG_HASH_VERSION,
    imported by the reference from a module that is not in its tree (:31)."""
    fields: List[IntLike] = [
        hash_version,
        config["max_funding_rate"],
        config["collateral_asset_info"]["asset_id"],
        config["collateral_asset_info"]["resolution"],
        config["fee_position_info"]["position_id"],
        config["fee_position_info"]["public_key"],
        config["positions_tree_height"],
        config["orders_tree_height"],
        config["timestamp_validation_config"]["price_validity_period"],
        config["timestamp_validation_config"]["funding_validity_period"],
        config["data_availability_mode"],
        config["is_risk_by_balance_only"],
    ]
    fields.append(str(len(fields)))
    return hash_chain_from_zero(fields).to_bytes(HASH_BYTES, "big")


def asset_hash(config: dict, asset_id: str, risk_upper_bound: int) -> bytes:
    """calculate_asset_hash (:125-175).  `risk_upper_bound` is RISK_UPPER_BOUND (:32, module absent
    from the reference tree)."""
    info = config["synthetic_assets_info"][asset_id]
    segments = info["risk_factor"]["segments"]
    fields: List[IntLike] = [asset_id, info["resolution"], len(segments)]
    fields += [seg["upper_bound"] * risk_upper_bound + int(seg["risk"]) for seg in segments]
    fields.append(len(info["oracle_price_signed_asset_ids"]))
    fields += info["oracle_price_signed_asset_ids"]
    fields.append(info["oracle_price_quorum"])
    fields.append(len(info["oracle_price_signers"]))
    fields += info["oracle_price_signers"]
    fields.append(str(len(fields)))
    return hash_chain_from_zero(fields).to_bytes(HASH_BYTES, "big")


def compute_hash_chain(data: Sequence[int]) -> int:
    """H(d0, H(d1, ... H(d_{n-2}, d_{n-1}))) - public cairo-lang `compute_hash_chain`."""
    return batch.pedersen_chain_right(list(data))


def program_hash_chain(program_data: Sequence[int], main: int, builtins: Sequence[int],
                       bootloader_version: int = 0) -> int:
    """Public cairo-lang `compute_program_hash_chain`: hash chain over
    [len(rest), bootloader_version, main, n_builtins, *builtins, *data]."""
    rest = [bootloader_version, main, len(builtins)] + list(builtins) + list(program_data)
    return compute_hash_chain([len(rest)] + rest)

"""The reference's own harness tests, restated against the MI355X backend:
services/perpetual/public/stark_cli_test.py:43-146 (CLI == in-process results, illegal parameters
produce stderr) plus the serial hash-chain consumers.  KAT data: tests/golden/reference_kats.json
(copied from the reference's fixtures)."""
import json
import os
import subprocess
import sys

import pytest

from oracle import ref_py as R

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.path.join(ROOT, "stark-perpetual_amd", "services", "perpetual", "public", "stark_cli.py")
KATS = json.load(open(os.path.join(ROOT, "tests", "golden", "reference_kats.json")))


def run_cli(*args):
    # the CLI picks its own small window plan; do not let a plan chosen for this test process leak
    # into the child (two 155 GiB tables would not fit one GPU)
    env = {k: v for k, v in os.environ.items() if k != "STARKPERP_WINDOW_BITS"}
    return subprocess.run([sys.executable, CLI] + list(args), capture_output=True, env=env)


HASH_ARGS = ["--oracle", "4d616b6572", "--asset", "42544355534400000000000000000000", "--price",
             "000000000000000000000000000000000000000000000000ac9f3163ad52b000", "--time",
             "000000000000000000000000000000000000000000000000000000005f590c1e"]


def test_cli_hash_args():
    from starkware.crypto.signature.signature import pedersen_hash
    out = run_cli("--method", "hash", *HASH_ARGS)
    assert out.stderr == b""
    exp = pedersen_hash(0x425443555344000000000000000000004D616B6572, 0xAC9F3163AD52B0005F590C1E)
    assert exp == int(KATS["stark_cli_hash"]["out"], 16)
    assert out.stdout == bytes(hex(exp)[2:] + "\n", "utf-8")


@pytest.mark.parametrize("idx,value", [
    (1, "14d616b6572"),
    (7, "000000000000000000000000000000000000000000000000000000015f590c1e"),
    (3, "4254435553440000000000000000000000"),
    (5, "000010000000000000000000000000000000000000000000ac9f3163ad52b000"),
])
def test_cli_hash_illegal_params(idx, value):
    args = list(HASH_ARGS)
    args[idx] = value
    out = run_cli("--method", "hash", *args)
    assert out.stderr != b""


def test_cli_sign():
    from starkware.crypto.signature.signature import sign
    a = KATS["party_a_order"]
    r, s = sign(int(a["message_hash"], 16), int(a["private_key"], 16))
    assert (hex(r), hex(s)) == (a["signature"]["r"], a["signature"]["s"])
    out = run_cli("--method", "sign", "--key", a["private_key"], "--data", a["message_hash"])
    assert out.stderr == b""
    assert out.stdout == bytes(" ".join([hex(r), hex(s)]) + "\n", "utf-8")


def test_public_key():
    private, public = list(KATS["keys_precomputed"].items())[0]
    out = run_cli("--method", "get_public", "--key", private)
    assert out.stderr == b""
    assert out.stdout == bytes(public + "\n", "utf-8")


def test_hash_chains_match_oracle():
    from starkperp import hash_chains as hc
    words = [5, 2**250 + 3, 0, R.FIELD_PRIME - 1, 77]
    acc = 0
    for w in words:
        acc = R.pedersen_hash(acc, w)
    assert hc.hash_chain_from_zero(words) == acc
    right = words[-1]
    for w in reversed(words[:-1]):
        right = R.pedersen_hash(w, right)
    assert hc.compute_hash_chain(words) == right
    assert hc.compute_hash_chain([9]) == 9
    cfg = {
        "max_funding_rate": 1120, "collateral_asset_info": {"asset_id": "0x2a", "resolution": "0xf4240"},
        "fee_position_info": {"position_id": 7, "public_key": "0x1ef15c18599971b7beced415a40f0c7deacfd9b0d1819e03d723d8bc943cfca"},
        "positions_tree_height": 64, "orders_tree_height": 64,
        "timestamp_validation_config": {"price_validity_period": 31536000, "funding_validity_period": 604800},
        "data_availability_mode": 0, "is_risk_by_balance_only": True,
        "synthetic_assets_info": {"0x4254432d3130000000000000000000": {
            "resolution": "0x2540be400", "risk_factor": {"segments": [{"upper_bound": 5, "risk": "214748365"}]},
            "oracle_price_signed_asset_ids": ["0x11", "0x12"], "oracle_price_quorum": 1,
            "oracle_price_signers": ["0x13"]}},
    }
    fields = [3, 1120, 0x2A, 0xF4240, 7, int(cfg["fee_position_info"]["public_key"], 16), 64, 64, 31536000,
              604800, 0, 1, 12]
    acc = 0
    for w in fields:
        acc = R.pedersen_hash(acc, w)
    assert hc.general_config_hash(cfg, 3) == acc.to_bytes(32, "big")
    aid = "0x4254432d3130000000000000000000"
    fields = [int(aid, 16), 0x2540BE400, 1, 5 * 2**32 + 214748365, 2, 0x11, 0x12, 1, 1, 0x13, 10]
    acc = 0
    for w in fields:
        acc = R.pedersen_hash(acc, w)
    assert hc.asset_hash(cfg, aid, 2**32) == acc.to_bytes(32, "big")
    rest = [0, 12, 2, 100, 200, 1, 2, 3]
    chain = [len(rest)] + rest
    exp = chain[-1]
    for w in reversed(chain[:-1]):
        exp = R.pedersen_hash(w, exp)
    assert hc.program_hash_chain([1, 2, 3], main=12, builtins=[100, 200]) == exp


def test_fast_pedersen_hash_overlay():
    from starkware.crypto.signature import fast_pedersen_hash as f
    k = KATS["hash_test"]["pedersen_hash_data_1"]
    x, y, o = (int(k[n], 16) for n in ("input_1", "input_2", "output"))
    assert f.pedersen_hash(x, y) == o
    assert f.pedersen_hash_func(x.to_bytes(32, "big"), y.to_bytes(32, "big")) == o.to_bytes(32, "big")
    with pytest.raises(AssertionError):
        f.pedersen_hash_func(b"\x00" * 31, b"\x00" * 32)

#!/usr/bin/env python3
"""
Adds tests/golden/g8_reference_fixtures_extra.json: the signature fixtures of the reference's
signature_test_data.json that g1-g7 did not use yet, run through the REAL reference
(same recipe as gen_golden.py):

    PYTHONPATH=/tmp/oracle_shim:/root/reference/src python3 oracle/gen_golden_extra.py

* every (message_hash, r, s, public_key) in the file with the reference's verify() verdict;
* for every meta_data entry that carries a private key: the reference's deterministic
  sign(message_hash, private_key) and private_to_stark_key(private_key).
"""
import json
import os

from starkware.crypto.signature import signature as ref

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = "/root/reference/src/starkware/crypto/signature/test/config/signature_test_data.json"


def main():
    data = json.load(open(SRC))
    md = data["meta_data"]
    orders = {"party_a_order": data["settlement"]["party_a_order"],
              "party_b_order": data["settlement"]["party_b_order"]}
    for name in ("transfer_order", "conditional_transfer_order", "transfer_order_2nd_valid_range",
                 "order_with_vault_id_in_2nd_range", "multi_asset_order"):
        orders[name] = data[name]
    verify_cases = {}
    for name, order in orders.items():
        z = int(md[name]["message_hash"], 16)
        r, s = int(order["signature"]["r"], 16), int(order["signature"]["s"], 16)
        pub = order.get("public_key")
        if pub is None:
            pub = hex(ref.private_to_stark_key(int(md[name]["private_key"], 16)))
        verify_cases[name] = {"message_hash": hex(z), "r": hex(r), "s": hex(s), "public_key": pub,
                              "reference_verify": bool(ref.verify(z, r, s, int(pub, 16)))}
    sign_cases = {}
    for name, entry in md.items():
        if "private_key" not in entry:
            continue
        z, d = int(entry["message_hash"], 16), int(entry["private_key"], 16)
        r, s = ref.sign(z, d)
        sign_cases[name] = {"message_hash": hex(z), "private_key": hex(d), "r": hex(r), "s": hex(s),
                            "public_key": hex(ref.private_to_stark_key(d)),
                            "reference_verify": bool(ref.verify(z, r, s, ref.private_to_stark_key(d)))}
    out = {"_source": "signature_test_data.json through the reference (oracle/gen_golden_extra.py)",
           "verify": verify_cases, "sign": sign_cases}
    with open(os.path.join(ROOT, "tests", "golden", "g8_reference_fixtures_extra.json"), "w") as f:
        json.dump(out, f, indent=1)
    print("verify cases:", {k: v["reference_verify"] for k, v in verify_cases.items()})
    print("sign cases:", len(sign_cases))


if __name__ == "__main__":
    main()

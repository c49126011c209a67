#!/usr/bin/env python3
"""
Generates tests/golden/*.json by running the REAL reference (imported from /root/reference/src).
Runs only in the build container (the reference does not travel to the GPU box); the JSON it
writes is data - inputs and expected outputs - and is committed.

Recipe: put scratch shims for `ecdsa.rfc6979.generate_k`, the moved
`sympy.core.numbers.igcdex` and an empty `web3` in a scratch dir (never in this repo), then

    PYTHONPATH=/tmp/oracle_shim:/root/reference/src python3 oracle/gen_golden.py [--heavy]

--heavy additionally produces the 2^16-leaf tree (config C2) and the 4096-order batch (config C3)
fixtures, which take several minutes on 8 cores.
"""

import argparse
import json
import multiprocessing as mp
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))

import workloads as wl  # noqa: E402

from starkware.crypto.signature import signature as ref  # noqa: E402
from services.perpetual.public import perpetual_messages as ref_msgs  # noqa: E402

REF = "/root/reference/src"
GOLD = os.path.join(ROOT, "tests", "golden")


def hx(v):
    return hex(v)


def dump(name, obj):
    with open(os.path.join(GOLD, name), "w") as f:
        json.dump(obj, f, indent=1)
        f.write("\n")
    print("wrote", name)


# ---- worker functions (module level for multiprocessing) ----
def _hash_pair(xy):
    return ref.pedersen_hash(xy[0], xy[1])


def _sign_case(c):
    z, d, sd = c
    return ref.sign(z, d, sd)


def _verify_case(c):
    z, r, s, q = c
    if isinstance(q, list):
        q = tuple(q)
    try:
        return "true" if ref.verify(z, r, s, q) else "false"
    except AssertionError as e:
        msg = str(e)
        return "assert:" + msg.split(" ")[0] if msg else "assert:"


def _pubkey(d):
    return ref.private_key_to_ec_point_on_stark_curve(d)


def _order_msg(args):
    return ref_msgs.get_limit_order_msg(*args)


def signature_fixtures(sig_data):
    """(message_hash, r, s, public_key) tuples held by signature_test_data.json, each with the
    reference's own verify() verdict (some of those JS-side fixtures are stale and verify False)."""
    md = sig_data["meta_data"]
    pairs = {
        "party_a_order": sig_data["settlement"]["party_a_order"],
        "party_b_order": sig_data["settlement"]["party_b_order"],
        "transfer_order": sig_data["transfer_order"],
        "conditional_transfer_order": sig_data["conditional_transfer_order"],
        "multi_asset_order": sig_data["multi_asset_order"],
    }
    out = {}
    for name, order in pairs.items():
        z = int(md[name]["message_hash"], 16)
        r, s = int(order["signature"]["r"], 16), int(order["signature"]["s"], 16)
        pub = order.get("public_key")
        if pub is None:
            pub = hex(ref.private_to_stark_key(int(md[name]["private_key"], 16)))
        out[name] = {"message_hash": hex(z), "r": hex(r), "s": hex(s), "public_key": pub,
                     "reference_verify": bool(ref.verify(z, r, s, int(pub, 16)))}
    return out


def ref_position_hash(pos):
    """position/hash.cairo:22-74 evaluated with the reference's pedersen_hash."""
    public_key, collateral, assets = pos
    acc = 0
    for asset_id, funding, balance in assets:
        packed = (asset_id * 2**64 + (funding + 2**63)) * 2**64 + (balance + 2**63)
        acc = ref.pedersen_hash(acc, packed)
    acc = ref.pedersen_hash(acc, public_key)
    return ref.pedersen_hash(acc, (collateral + 2**63) * 2**16 + len(assets))


def tree_levels(pool, leaves):
    levels = [leaves]
    while len(levels[-1]) > 1:
        cur = levels[-1]
        pairs = [(cur[2 * i], cur[2 * i + 1]) for i in range(len(cur) // 2)]
        levels.append(pool.map(_hash_pair, pairs, chunksize=max(1, len(pairs) // 64)))
    return levels


def sparse_update_root(pool, height, mods, empties):
    layer = dict(mods)
    for level in range(height):
        parents = sorted(set(i // 2 for i in layer))
        pairs = [
            (layer.get(2 * i, empties[level]), layer.get(2 * i + 1, empties[level]))
            for i in parents
        ]
        vals = pool.map(_hash_pair, pairs, chunksize=max(1, len(pairs) // 64))
        layer = dict(zip(parents, vals))
    return layer[0]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--heavy", action="store_true")
    ap.add_argument("--only-heavy", action="store_true")
    args = ap.parse_args()
    os.makedirs(GOLD, exist_ok=True)
    pool = mp.Pool(8)
    P, N = ref.FIELD_PRIME, ref.EC_ORDER

    if not args.only_heavy:
        # ---- reference-held vectors, copied as data ----
        sig_data = json.load(
            open(f"{REF}/starkware/crypto/signature/test/config/signature_test_data.json"))
        keys = json.load(
            open(f"{REF}/starkware/crypto/signature/src/config/keys_precomputed.json"))
        msgs = json.load(open(f"{REF}/services/perpetual/public/perpetual_messages_precomputed.json"))
        dump("reference_kats.json", {
            "_source": "data copied from the reference's own test fixtures: "
                       "signature_test_data.json (hash_test, party_a_order), "
                       "keys_precomputed.json, perpetual_messages_precomputed.json",
            "hash_test": sig_data["hash_test"],
            "party_a_order": {
                "message_hash": sig_data["meta_data"]["party_a_order"]["message_hash"],
                "private_key": sig_data["meta_data"]["party_a_order"]["private_key"],
                "public_key": sig_data["settlement"]["party_a_order"]["public_key"],
                "signature": sig_data["settlement"]["party_a_order"]["signature"],
            },
            "signature_fixtures": signature_fixtures(sig_data),
            "keys_precomputed": keys,
            "perpetual_messages": msgs,
            "stark_cli_hash": {
                "x": "0x425443555344000000000000000000004D616B6572",
                "y": "0xAC9F3163AD52B0005F590C1E",
                "out": hx(ref.pedersen_hash(0x425443555344000000000000000000004D616B6572,
                                            0xAC9F3163AD52B0005F590C1E)),
            },
        })

        # ---- parameters ----
        import hashlib
        h = hashlib.sha256()
        for x, y in ref.CONSTANT_POINTS:
            h.update(x.to_bytes(32, "big") + y.to_bytes(32, "big"))
        dump("params_digest.json", {
            "FIELD_PRIME": hx(P), "EC_ORDER": hx(N), "ALPHA": ref.ALPHA, "BETA": hx(ref.BETA),
            "FIELD_GEN": ref.FIELD_GEN, "n_points": len(ref.CONSTANT_POINTS),
            "constant_points_sha256": h.hexdigest(),
            "sample_points": {str(i): [hx(ref.CONSTANT_POINTS[i][0]), hx(ref.CONSTANT_POINTS[i][1])]
                              for i in (0, 1, 2, 3, 249, 250, 253, 254, 501, 502, 505)},
        })

        # ---- G1: Pedersen ----
        pairs = wl.pedersen_pairs(1024, seed=0)
        outs = pool.map(_hash_pair, pairs, chunksize=16)
        edges = wl.edge_pairs()
        edge_out = pool.map(_hash_pair, edges, chunksize=2)
        dump("g1_pedersen.json", {
            "seed": 0, "n": 1024, "digest": wl.digest_felts(outs),
            "first16": [hx(v) for v in outs[:16]],
            "all": [hx(v) for v in outs],
            "edge": [[hx(a), hx(b), hx(o)] for (a, b), o in zip(edges, edge_out)],
            "arity": {"zero": hx(ref.pedersen_hash()), "one_1": hx(ref.pedersen_hash(1)),
                      "one_pm1": hx(ref.pedersen_hash(P - 1)),
                      "point_1_2": [hx(v) for v in ref.pedersen_hash_as_point(1, 2)]},
        })

        # ---- G2: keys ----
        ds = wl.private_keys(256, seed=10)
        pubs = pool.map(_pubkey, ds, chunksize=8)
        dump("g2_keys.json", {"seed": 10, "keys": [[hx(d), hx(q[0]), hx(q[1])]
                                                  for d, q in zip(ds, pubs)]})

        # ---- G3: sign ----
        cases = wl.sign_cases(256, seed=11)
        sigs = pool.map(_sign_case, cases, chunksize=8)
        ks = [ref.generate_k_rfc6979(z, d, sd) for z, d, sd in cases]
        dump("g3_sign.json", {"seed": 11, "cases": [
            [hx(z), hx(d), (None if sd is None else hx(sd)), hx(r), hx(s), hx(k)]
            for (z, d, sd), (r, s), k in zip(cases, sigs, ks)]})

        # ---- G4: verify ----
        vcases, labels = [], []
        pub_of = {d: q for d, q in zip(ds, pubs)}
        for i, ((z, d, sd), (r, s)) in enumerate(zip(cases[:96], sigs[:96])):
            d_pub = ref.private_key_to_ec_point_on_stark_curve(d)
            kind = i % 6
            if kind == 0:
                vcases.append((z, r, s, list(d_pub))); labels.append("valid_point")
            elif kind == 1:
                vcases.append((z, r, s, d_pub[0])); labels.append("valid_xonly")
            elif kind == 2:
                vcases.append(((z + 1) % 2**251, r, s, d_pub[0])); labels.append("wrong_z")
            elif kind == 3:
                vcases.append((z, (r % (2**251 - 1)) + 1, s, list(d_pub))); labels.append("wrong_r")
            elif kind == 4:
                vcases.append((z, r, (s % (N - 1)) + 1, d_pub[0])); labels.append("wrong_s")
            else:
                other = pubs[(i * 7 + 3) % len(pubs)]
                vcases.append((z, r, s, list(other))); labels.append("wrong_key")
        # negated-y point key still verifies (r only depends on x through +/-): both branches
        z, d, sd = cases[40]
        r, s = sigs[40]
        q = ref.private_key_to_ec_point_on_stark_curve(d)
        vcases.append((z, r, s, [q[0], P - q[1]])); labels.append("neg_y_point")
        # pre-assert failures and AIR-style failures
        vcases.append((z, r, 0, list(q))); labels.append("s_zero")
        vcases.append((z, r, N, list(q))); labels.append("s_eq_N")
        vcases.append((z, 0, s, list(q))); labels.append("r_zero")
        vcases.append((z, 2**251, s, list(q))); labels.append("r_2p251")
        w_big = 2**251 + 5
        vcases.append((z, r, ref.inv_mod_curve_size(w_big), list(q))); labels.append("w_big")
        vcases.append((2**251, r, s, list(q))); labels.append("z_2p251")
        vcases.append((z, r, s, [q[0], (q[1] + 1) % P])); labels.append("off_curve_point")
        bad_x = 1
        while ref.is_valid_stark_key(bad_x):
            bad_x += 1
        vcases.append((z, r, s, bad_x)); labels.append("off_curve_xonly")
        # z = 0 is signable but the AIR ladder rejects m == 0
        r0, s0 = ref.sign(0, d)
        vcases.append((0, r0, s0, list(q))); labels.append("z_zero_point")
        vcases.append((0, r0, s0, q[0])); labels.append("z_zero_xonly")
        # public key equal to the shift point: x-collision at ladder step 0
        vcases.append((z, r, s, list(ref.SHIFT_POINT))); labels.append("key_is_shift")
        vcases.append((z, r, s, list(ref.EC_GEN))); labels.append("key_is_gen")
        # z*G + r*Q = infinity: z = -r*d mod N
        rr = r
        while True:
            zz = (-rr * d) % N
            if zz < 2**251:
                break
            rr += 1
        vcases.append((zz, rr, s, list(q))); labels.append("zG_plus_rQ_infinity")
        vcases.append((zz, rr, s, q[0])); labels.append("zG_plus_rQ_infinity_xonly")
        vouts = pool.map(_verify_case, vcases, chunksize=2)
        dump("g4_verify.json", {"cases": [
            {"label": lb, "z": hx(c[0]), "r": hx(c[1]), "s": hx(c[2]),
             "key": ([hx(c[3][0]), hx(c[3][1])] if isinstance(c[3], list) else hx(c[3])),
             "expect": o}
            for lb, c, o in zip(labels, vcases, vouts)]})

        # ---- G5: message hashes ----
        orders = wl.limit_orders(256, seed=2)
        zs = pool.map(_order_msg, [wl.order_args(o) for o in orders], chunksize=4)
        dump("g5_messages.json", {
            "seed": 2, "limit_order_z": [hx(v) for v in zs],
            "transfer": hx(ref_msgs.get_transfer_msg(5, 6, 7, 8, 9, 10, 11, 12, 13, 14)),
            "conditional_transfer": hx(
                ref_msgs.get_conditional_transfer_msg(5, 6, 7, 99, 8, 9, 10, 11, 12, 13, 14)),
            "withdrawal_to_address": hx(
                ref_msgs.get_withdrawal_to_address_msg(5, 6, "0xabcdef0123", 7, 8, 9)),
            "price": hx(ref_msgs.get_price_msg(0x4d616b6572, 0x42544355534400000000000000000000,
                                               0x5f590c1e, 0xac9f3163ad52b000)),
        })

        # ---- G6: Merkle (small) ----
        roots = {}
        for hgt in range(0, 11):
            lv = wl.leaves(1 << hgt, seed=100 + hgt)
            roots[str(hgt)] = hx(tree_levels(pool, lv)[-1][0])
        empties0 = [0]
        for _ in range(64):
            empties0.append(ref.pedersen_hash(empties0[-1], empties0[-1]))
        poss = wl.positions(64, seed=3)
        pos_hashes = pool.map(ref_position_hash, poss, chunksize=2)
        empty_pos = ref_position_hash((0, 0, []))
        # small sparse multi-update: 37 leaves in a height-20 tree of zeros
        import random
        rng = random.Random(77)
        mods = {rng.randrange(1 << 20): rng.randrange(P) for _ in range(37)}
        dump("g6_merkle.json", {
            "roots_seed_100_plus_h": roots,
            "empty_roots_leaf0": [hx(v) for v in empties0],
            "position_hashes_seed3": [hx(v) for v in pos_hashes],
            "empty_position_leaf": hx(empty_pos),
            "sparse_h20_seed77": {
                "mods": [[k, hx(v)] for k, v in sorted(mods.items())],
                "root": hx(sparse_update_root(pool, 20, mods, empties0)),
            },
        })

    if args.heavy or args.only_heavy:
        # ---- C2: 2^16-leaf rebuild ----
        lv = wl.leaves(1 << 16, seed=1)
        levels = tree_levels(pool, lv)
        dump("g6_c2_tree.json", {
            "seed": 1, "height": 16, "root": hx(levels[-1][0]),
            "left_spine": [hx(l[0]) for l in levels],
            "level_digests": [wl.digest_felts(l) for l in levels],
        })
        # ---- C3: 4096-order batch ----
        orders = wl.limit_orders(4096, seed=2)
        zs = pool.map(_order_msg, [wl.order_args(o) for o in orders], chunksize=16)
        keys = wl.private_keys(1024, seed=12)
        pubs = pool.map(_pubkey, keys, chunksize=16)
        # z must be < 2^251 to be signable; the (rare) others are signed on z mod 2^251
        scases = [(z % 2**251, keys[o["key_index"]], None) for z, o in zip(zs, orders)]
        sigs = pool.map(_sign_case, scases, chunksize=16)
        vcases = [(c[0], r, s, pubs[o["key_index"]][0])
                  for c, (r, s), o in zip(scases, sigs, orders)]
        # corrupt every 16th signature so the batch has both outcomes
        vcases = [(z, r, (s % (N - 1)) + 1, q) if i % 16 == 5 else (z, r, s, q)
                  for i, (z, r, s, q) in enumerate(vcases)]
        vouts = pool.map(_verify_case, vcases, chunksize=16)
        empties0 = [0]
        for _ in range(64):
            empties0.append(ref.pedersen_hash(empties0[-1], empties0[-1]))
        mods = {}
        for z, o in zip(zs, orders):
            order_id = z >> 187  # top 64 bits of the 251-bit message hash (order/order.cairo:23-59)
            mods[order_id] = o["amount_synthetic"]  # fulfilled amount felt (order.cairo:122-124)
        root64 = sparse_update_root(pool, 64, mods, empties0)
        dump("g7_c3_batch.json", {
            "orders_seed": 2, "keys_seed": 12, "n": 4096,
            "z_digest": wl.digest_felts(zs), "z_first4": [hx(v) for v in zs[:4]],
            "pub_digest": wl.digest_felts([q[0] for q in pubs]),
            "r_digest": wl.digest_felts([r for r, _ in sigs]),
            "s_digest": wl.digest_felts([s for _, s in sigs]),
            "verify_bits": "".join("1" if v == "true" else "0" for v in vouts),
            "orders_tree_height": 64, "orders_tree_root": hx(root64),
            "n_distinct_order_ids": len(mods),
        })
    pool.close()


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Reproducible record of the search behind "the multi_asset_order fixture is from another revision".

The reference holds ONE vector for the multi-asset order message (signature_test_data.json:102-139, hash at
:185-188; copied as data into tests/golden/reference_kats.json).  Its (hash, key, r, s) part reproduces; the
hash itself does not follow from the fixture's fields under signature_message_hashes.cairo:171-471 of this
tree.  This script enumerates layout variants around the Cairo source and hashes every one with the C oracle
(oracle/starkref.c, pinned pedersen_hash):

  stage A (structure)   list order (receive|give first) x index rule (own list / global, base 0 / 1) x who counts
                        as third party (key != signer | entry carries a key) x order of the four counts in the
                        metadata word (24) x order of the five word groups (120)              = 46 080 variants
  stage B (encoding)    group order (120) x list order x index rule x field order inside a packed felt (MSB / LSB
                        first) x (vault, amount) | (amount, vault) x padding shift 3 | 0 x left | right fold x
                        initial value words[0] | 0                                              = 30 720 variants
  stage C (order type)  the 10-bit order type 0 .. 31 (this tree: 6) x group order (120) x list order x index rule
                                                                                                = 30 720 variants

Prints the number of variants tried, whether any reproduces the fixture's hash, and the hash of the layout
that IS the Cairo source of this tree (the value tests/test_multi_asset_order.py pins).

    python tools/search_multi_asset_layout.py            # ~2 minutes on 8 cores
"""
import itertools
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import cref, ref_py  # noqa: E402

TYPE = ref_py.MULTI_ASSET_OFFCHAIN_ORDER_TYPE  # 6, signature_message_hashes.cairo


def load():
    fx = json.load(open(os.path.join(ROOT, "tests", "golden", "reference_kats.json")))["multi_asset_order"]
    key = int(fx["receive"][0]["public_key"], 16)

    def info(e):
        return (int(e["vault_id"]), int(e["public_key"], 16) if "public_key" in e else None, int(e["asset_id"], 16),
                int(e["amount"]))
    return fx, key, [info(e) for e in fx["give"]], [info(e) for e in fx["receive"]], [int(c, 16) for c in fx["conditions"]]


def words(fx, key, give, receive, conds, list_order, index_rule, third_rule, count_perm, group_perm, msb=True,
          va_order=True, shift=3, order_type=TYPE):
    lists = (receive, give) if list_order == 0 else (give, receive)
    va, assets, keys, idx = [], [], [], []
    g = 0
    for entries in lists:
        for i, (vault, pk, asset, amount) in enumerate(entries):
            assets.append(asset)
            va += [vault, amount] if va_order else [amount, vault]
            third = (pk is not None and pk != key) if third_rule == 0 else (pk is not None)
            if third:
                keys.append(pk)
                idx.append((i if index_rule < 2 else g) + (index_rule & 1))
            g += 1

    def pack(vals, per, bits):
        out = []
        for i in range(0, len(vals), per):
            acc = 0
            chunk = vals[i : i + per]
            for j, v in enumerate(chunk):
                acc = acc * 2**bits + v if msb else acc + (v << (bits * j))
            out.append(acc)
        return out
    groups = [list(conds), assets, keys, pack(va, 3, 64), pack(idx, 20, 12)]
    ws = []
    for gi in group_perm:
        ws += groups[gi]
    counts = [len(give), len(receive), len(idx), len(conds)]
    meta = order_type
    meta = meta * 2**32 + fx["nonce"]
    meta = meta * 2**32 + fx["expiration_timestamp"]
    for ci in count_perm:
        meta = meta * 2**12 + counts[ci]
    meta = meta * 2**126 + int(fx["system_id"], 16)
    ws.append(meta << shift)
    return ws


def fold_all(chains, left=True, init_first=True):
    """Hashes many word lists at once with the C oracle; returns one hash per chain."""
    if not left:
        chains = [list(reversed(c)) for c in chains]
    acc = [c[0] if init_first else 0 for c in chains]
    pos = [1 if init_first else 0] * len(chains)
    live = list(range(len(chains)))
    while live:
        xs, ys = [], []
        for k in live:
            a, w = acc[k], chains[k][pos[k]]
            xs.append(a if left else w)
            ys.append(w if left else a)
        hs = cref.opt_pedersen_hash_many(xs, ys)[0]
        nxt = []
        for k, h in zip(live, hs):
            acc[k] = h
            pos[k] += 1
            if pos[k] < len(chains[k]):
                nxt.append(k)
        live = nxt
    return acc


def main():
    fx, key, give, receive, conds = load()
    target = int(fx["message_hash"], 16)
    P = 2**251 + 17 * 2**192 + 1
    tried, hits = 0, []
    cairo = fold_all([words(fx, key, give, receive, conds, 0, 0, 0, (0, 1, 2, 3), (0, 1, 2, 3, 4))])[0]
    print("layout of signature_message_hashes.cairo:387-471 in this tree: 0x%x" % cairo)
    print("fixture (signature_test_data.json:185-188):                    0x%x" % target)
    # stage A
    labels, chains = [], []
    for lo, ir, tr in itertools.product(range(2), range(4), range(2)):
        for cp in itertools.permutations(range(4)):
            for gp in itertools.permutations(range(5)):
                labels.append(("A", lo, ir, tr, cp, gp))
                chains.append(words(fx, key, give, receive, conds, lo, ir, tr, cp, gp))
    got = fold_all(chains)
    tried += len(chains)
    hits += [l for l, h in zip(labels, got) if h == target]
    print("stage A: %d structural variants, %d reproduce the fixture" % (len(chains), len([h for h in got if h == target])))
    # stage B
    for left, init_first in itertools.product((True, False), (True, False)):
        labels, chains = [], []
        for gp in itertools.permutations(range(5)):
            for lo, ir, msb, vo, sh in itertools.product(range(2), range(4), (True, False), (True, False), (3, 0)):
                ws = words(fx, key, give, receive, conds, lo, ir, 0, (0, 1, 2, 3), gp, msb, vo, sh)
                if any(w >= P for w in ws):
                    continue
                labels.append(("B", left, init_first, gp, lo, ir, msb, vo, sh))
                chains.append(ws)
        got = fold_all(chains, left, init_first)
        tried += len(chains)
        hits += [l for l, h in zip(labels, got) if h == target]
        print("stage B (left fold %s, initial value words[0] %s): %d variants, %d reproduce the fixture"
              % (left, init_first, len(chains), len([h for h in got if h == target])))
    labels, chains = [], []
    for ot in range(32):
        for gp in itertools.permutations(range(5)):
            for lo, ir in itertools.product(range(2), range(4)):
                labels.append(("C", ot, gp, lo, ir))
                chains.append(words(fx, key, give, receive, conds, lo, ir, 0, (0, 1, 2, 3), gp, order_type=ot))
    got = fold_all(chains)
    tried += len(chains)
    hits += [l for l, h in zip(labels, got) if h == target]
    print("stage C: %d order-type variants, %d reproduce the fixture" % (len(chains), len([h for h in got if h == target])))
    print("total: %d variants tried, %d reproduce the fixture's message hash" % (tried, len(hits)))
    for h in hits:
        print("HIT", h)


if __name__ == "__main__":
    main()

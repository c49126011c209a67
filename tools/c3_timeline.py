#!/usr/bin/env python3
"""Timeline of ONE sp_order_batch call from a rocprofv3 kernel trace of `tools/c3_probe.py one`:

    python tools/c3_timeline.py <kernel_trace.csv>

Takes the kernels after the last gap of more than 100 ms (the probe sleeps 300 ms in front of the traced call),
prints start / duration / queue / name, then: the span of the message-hash chains, of the verification (its own
stream) and of the tree update (the tree's stream), how much of the verification overlaps the tree's levels, the
number of dependent hash launches of the update and the floor they imply."""
import csv
import sys


def main():
    rows = []
    for r in csv.DictReader(open(sys.argv[1])):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", ""),
                     r.get("Queue_Id", "?"), r.get("Stream_Id", r.get("Queue_Id", "?"))))
    rows.sort()
    cut = 0
    for i in range(1, len(rows)):
        if rows[i][0] - max(r[1] for r in rows[:i][-50:]) > 100e6:
            cut = i
    call = rows[cut:]
    t0 = call[0][0]
    print("%d kernels in the traced call (after a %.0f ms gap)" % (len(call), (call[0][0] - rows[cut - 1][1]) / 1e6 if cut else 0))
    for s, e, name, q, st in call:
        print("%9.1f us  dur %7.1f  q %-3s  %s" % ((s - t0) / 1e3, (e - s) / 1e3, q, name[:90]))

    def span(pred):
        sel = [(s, e) for s, e, n, q, st in call if pred(n)]
        if not sel:
            return None
        return min(s for s, _ in sel), max(e for _, e in sel), sum(e - s for s, e in sel), len(sel)
    chains = span(lambda n: "ped_chain" in n)
    verify = span(lambda n: "ecdsa_verify" in n)
    tree_hash = span(lambda n: "ped_path" in n or "ped_quad" in n or "ped_accumulate" in n or "ped_finish" in n or "ped_top" in n
                     or "ped_split" in n)
    tree_all = span(lambda n: n.startswith("sp::tree_") or "ped_path" in n or "ped_quad" in n)
    print()
    for label, sp in (("message-hash chains (ped_chain_kernel)", chains), ("verification (ecdsa_verify_keyed_kernel)", verify),
                      ("tree update: hash launches", tree_hash), ("tree update: all kernels", tree_all)):
        if sp:
            print("%-44s %8.1f .. %8.1f us  busy %8.1f us in %d launches" % (
                label, (sp[0] - t0) / 1e3, (sp[1] - t0) / 1e3, sp[2] / 1e3, sp[3]))
    if verify and tree_hash:
        ov = max(0, min(verify[1], tree_hash[1]) - max(verify[0], tree_hash[0]))
        print("verification overlaps the tree's hash launches for %.1f us of its %.1f us" % (ov / 1e3, (verify[1] - verify[0]) / 1e3))
    print("device span of the call: %.1f us" % ((max(e for _, e, *_ in call) - t0) / 1e3))


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Hash / sign / get_public command line with the flags and output format of the reference's
services/perpetual/public/stark_cli.py:47-181 (the harness behind stark_cli_test.py), running on
the MI355X backend.

    stark_cli.py --method hash --oracle 4d616b6572 --asset 4254...00 --price ...b000 --time ...0c1e
    stark_cli.py --method sign --key <hex private key> --data <hex message hash>
    stark_cli.py --method get_public --key <hex private key>
"""
import argparse
import os
import sys
import traceback

# the CLI never uses torch: stay on the system HIP runtime (keeps stderr empty, see starkperp._lib)
os.environ.setdefault("STARKPERP_SKIP_TORCH_RUNTIME", "1")
# one call per process: 2^12-entry windows (39 table entries per hash, 10 MiB, built in milliseconds)
# instead of the 4.3 GiB default that a long-lived service amortises
os.environ.setdefault("STARKPERP_WINDOW_BITS", "12")

_HERE = os.path.dirname(os.path.abspath(__file__))
_PKG_ROOT = os.path.normpath(os.path.join(_HERE, "..", "..", ".."))  # .../stark-perpetual_amd
if _PKG_ROOT not in sys.path:
    sys.path.insert(0, _PKG_ROOT)

from starkperp.perpetual_messages import get_price_msg  # noqa: E402
from starkperp.signature import EC_ORDER, FIELD_PRIME, private_to_stark_key, sign  # noqa: E402


def bounded_hex(bound):
    def parse(text):
        value = int(text, 16)
        assert value < bound
        return value
    return parse


def run(argv):
    top = argparse.ArgumentParser(description="Starkware hash & sign cli (MI355X backend).")
    top.add_argument("-m", "--method", required=True, dest="method",
                     choices=["hash", "sign", "get_public"],
                     help="The required operation - hash, sign or get_public")
    args, rest = top.parse_known_args(argv)
    sub = argparse.ArgumentParser()
    if args.method == "hash":
        sub.add_argument("-a", "--asset", required=True, type=bounded_hex(2**128), help="The asset pair")
        sub.add_argument("-o", "--oracle", required=True, type=bounded_hex(2**40), help="The signing oracle")
        sub.add_argument("-p", "--price", required=True, type=bounded_hex(2**120), help="The asset price")
        sub.add_argument("-t", "--time", required=True, type=bounded_hex(2**32), help="The asset time")
        ns = sub.parse_args(rest)
        return hex(get_price_msg(ns.oracle, ns.asset, ns.time, ns.price))[2:]
    # The reference bounds --key by FIELD_PRIME (stark_cli.py:59-61); keys in [EC_ORDER, FIELD_PRIME) are not
    # valid Stark private keys (signature.py:197-201) and the library rejects them, so the parser does too.
    sub.add_argument("-k", "--key", required=True, type=bounded_hex(EC_ORDER),
                     help="The private key (hex string)")
    if args.method == "sign":
        sub.add_argument("-d", "--data", required=True, type=bounded_hex(FIELD_PRIME),
                         help="The data to sign")
        ns = sub.parse_args(rest)
        r, s = sign(ns.data, ns.key)
        return " ".join([hex(r), hex(s)])
    ns = sub.parse_args(rest)
    return hex(private_to_stark_key(ns.key))


def main():
    try:
        print(run(sys.argv[1:]))
        return 0
    except Exception:
        print('Got an error while processing "%s":' % sys.argv[0], file=sys.stderr)
        traceback.print_exc()
        print(file=sys.stderr)
        return 1


if __name__ == "__main__":
    sys.exit(main())

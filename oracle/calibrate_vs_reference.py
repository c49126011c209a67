#!/usr/bin/env python3
"""
Speed calibration of the CPU "port" baseline against the REAL reference:
the same 256 Pedersen hashes + 64 signatures + 64 verifications (C1 inputs, random.Random(0)) through
/root/reference/src/starkware/crypto/signature/signature.py and through oracle/ref_py.py, one core each,
outputs compared item by item, the time ratio printed.  bench.py's `cpu_baseline` legs time ref_py.py on the
GPU box (the reference cannot travel); dividing them by the ratios printed here reads them as
"reference-equivalent".

Test infrastructure: runs only in the build container (it imports the reference).  The scratch shims of
SURVEY.md Appendix A (ecdsa.rfc6979.generate_k, the moved sympy igcdex, an empty web3) are written to a
temporary directory, never into this repo.

    python3 oracle/calibrate_vs_reference.py [--out profiles/r03_cpu_calibration.txt]
"""

import argparse
import os
import random
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_SRC = "/root/reference/src"

SHIM_RFC6979 = '''
import hmac


def _bits2int(b, qlen):
    v = int.from_bytes(b, "big")
    extra = len(b) * 8 - qlen
    return v >> extra if extra > 0 else v


def generate_k(order, secexp, hash_func, data, retry_gen=0, extra_entropy=b""):
    qlen = order.bit_length()
    holen = hash_func().digest_size
    rolen = (qlen + 7) // 8
    olen = (len("%x" % order) + 1) // 2
    z = _bits2int(data, qlen)
    if z >= order:
        z -= order
    seed = secexp.to_bytes(olen, "big") + z.to_bytes(olen, "big") + extra_entropy
    mac = lambda key, msg: hmac.new(key, msg, hash_func).digest()
    v = b"\\x01" * holen
    k = b"\\x00" * holen
    k = mac(k, v + b"\\x00" + seed)
    v = mac(k, v)
    k = mac(k, v + b"\\x01" + seed)
    v = mac(k, v)
    while True:
        t = b""
        while len(t) < rolen:
            v = mac(k, v)
            t += v
        cand = _bits2int(t, qlen)
        if 1 <= cand < order:
            if retry_gen <= 0:
                return cand
            retry_gen -= 1
        k = mac(k, v + b"\\x00")
        v = mac(k, v)
'''

SHIM_SITE = '''
import sympy.core.numbers as _n
if not hasattr(_n, "igcdex"):
    from sympy.core.intfunc import igcdex as _g
    _n.igcdex = _g
'''

SHIM_WEB3 = '''
class Web3:
    @staticmethod
    def solidityKeccak(*a, **k):
        raise NotImplementedError("web3 is not installed")


class HTTPProvider:
    pass
'''


def c1_inputs():
    """C1 of BASELINE.json: 256 hash pairs, 64 (z, d) pairs from random.Random(0)."""
    rng = random.Random(0)
    p = 2**251 + 17 * 2**192 + 1
    n = 0x0800000000000010FFFFFFFFFFFFFFFFB781126DCAE7B2321E66A241ADC64D2F
    pairs = [(rng.randrange(p), rng.randrange(p)) for _ in range(256)]
    zd = [(rng.randrange(2**251), rng.randrange(1, n)) for _ in range(64)]
    return pairs, zd


def run_legs(which):
    """Child process: time the three legs through one implementation, print 'name seconds digest'."""
    import hashlib

    if which == "reference":
        from starkware.crypto.signature import signature as impl
    else:
        sys.path.insert(0, ROOT)
        from oracle import ref_py as impl
    pairs, zd = c1_inputs()
    impl.pedersen_hash(1, 2)  # tables / imports outside the clock
    t0 = time.perf_counter()
    hs = [impl.pedersen_hash(x, y) for x, y in pairs]
    t1 = time.perf_counter()
    sigs = [impl.sign(z, d) for z, d in zd]
    t2 = time.perf_counter()
    keys = [impl.private_to_stark_key(d) for _, d in zd]
    t3 = time.perf_counter()
    oks = [impl.verify(z, r, s, q) for (z, _), (r, s), q in zip(zd, sigs, keys)]
    t4 = time.perf_counter()
    dig = lambda v: hashlib.sha256(repr(v).encode()).hexdigest()[:16]
    print("hash", t1 - t0, dig(hs))
    print("sign", t2 - t1, dig(sigs))
    print("verify", t4 - t3, dig(oks) + ("" if all(oks) else "-NOT-ALL-TRUE"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--leg", choices=["reference", "port"])
    ap.add_argument("--out")
    args = ap.parse_args()
    if args.leg:
        run_legs(args.leg)
        return
    if not os.path.isdir(REF_SRC):
        sys.exit("the reference tree is not present: this script runs in the build container only")
    res = {}
    with tempfile.TemporaryDirectory(prefix="oracle_shim_") as shim:
        os.makedirs(os.path.join(shim, "ecdsa"))
        os.makedirs(os.path.join(shim, "web3"))
        open(os.path.join(shim, "ecdsa", "__init__.py"), "w").close()
        open(os.path.join(shim, "ecdsa", "rfc6979.py"), "w").write(SHIM_RFC6979)
        open(os.path.join(shim, "sitecustomize.py"), "w").write(SHIM_SITE)
        open(os.path.join(shim, "web3", "__init__.py"), "w").write(SHIM_WEB3)
        for leg in ("reference", "port"):
            env = dict(os.environ)
            env["PYTHONPATH"] = shim + ":" + REF_SRC if leg == "reference" else ""
            env["PYTHONDONTWRITEBYTECODE"] = "1"
            out = subprocess.run([sys.executable, os.path.abspath(__file__), "--leg", leg], env=env, check=True,
                                 capture_output=True, text=True).stdout
            res[leg] = {ln.split()[0]: (float(ln.split()[1]), ln.split()[2]) for ln in out.splitlines() if ln.strip()}
    counts = {"hash": 256, "sign": 64, "verify": 64}
    lines = ["CPU calibration: oracle/ref_py.py (the 'port' baseline of bench.py) against the reference itself",
             "(/root/reference/src/starkware/crypto/signature/signature.py:137-173, 217-260, 296-318), one core each,",
             "C1 inputs of BASELINE.json (256 hash pairs, 64 (z, d) pairs, random.Random(0)); build container, "
             + str(os.cpu_count()) + " vCPUs.", "",
             "leg      items   reference s   ms/item     port s   ms/item   reference/port   outputs equal"]
    for leg in ("hash", "sign", "verify"):
        (tr, dr), (tp, dp) = res["reference"][leg], res["port"][leg]
        n = counts[leg]
        lines.append("%-7s %6d   %11.3f %9.3f %10.3f %9.3f %16.2f   %s"
                     % (leg, n, tr, 1e3 * tr / n, tp, 1e3 * tp / n, tr / tp, dr == dp))
    lines += ["",
              "Reading: a `cpu_baseline` figure of bench.py (ref_py.py on the GPU box) divided by the ratio of its leg is",
              "the reference-equivalent rate on the same cores.  The port is faster because its modular inverse is a plain",
              "extended Euclid on Python ints where the reference calls sympy's igcdex (math_utils.py:50-55)."]
    text = "\n".join(lines) + "\n"
    print(text, end="")
    if args.out:
        with open(args.out, "w") as f:
            f.write(text)
    if not all(res["reference"][k][1] == res["port"][k][1] for k in counts):
        sys.exit("outputs differ")


if __name__ == "__main__":
    main()

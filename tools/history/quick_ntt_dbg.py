import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "stark-perpetual_amd")); sys.path.insert(0, ROOT)
import torch, random
from oracle import stark_ref as S
from starkperp import stark
P = S.P
for log_n in (9, 10, 12, 13):
    n = 1 << log_n
    rng = random.Random(log_n)
    c = [0] * n
    for k, v in {0: 5, 1: 7, 57: 11, n - 1: 13, n // 2 + 1: 17}.items(): c[k] = v
    t = stark.felts_to_tensor(c)
    ev = stark.ntt(t)
    back = stark.tensor_to_felts(stark.ntt(ev, inverse=True))
    diff = [i for i in range(n) if back[i] != c[i]]
    print(log_n, "mismatches", len(diff), diff[:8], [hex(back[i]) for i in diff[:3]])
    w = S.root_of_unity(log_n)
    ref = S.ntt(c, w)
    evl = stark.tensor_to_felts(ev)
    d2 = [i for i in range(n) if evl[i] != ref[i]]
    print("   forward mismatches", len(d2), d2[:8])
    c2 = [rng.randrange(P) for _ in range(n)]
    b2 = stark.tensor_to_felts(stark.ntt(stark.ntt(stark.felts_to_tensor(c2)), inverse=True))
    print("   random round trip mismatches", sum(1 for i in range(n) if b2[i] != c2[i]))

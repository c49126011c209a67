"""Constants and small helpers shared by bench.py and the benchlib modules."""
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEIGHT = 16
HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s spec
ALGO_BYTES_PER_HASH = 96  # two 32-byte felts in, one out
DTYPE = "u32x9"  # 29-bit limbs in 32-bit registers, 64-bit accumulators, full-width arithmetic mod p = 2^251 + 17*2^192 + 1


def seeded_felts(torch, n, seed, device):
    """n felts < 2^250 as an int64 [n, 4] tensor (little-endian limbs)."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    t = torch.randint(-(2**63), 2**63 - 1, (n, 4), dtype=torch.int64, generator=g)
    t[:, 3] &= (1 << 58) - 1
    return t.to(device)


def median(v):
    s = sorted(v)
    return s[len(s) // 2] if len(s) % 2 else 0.5 * (s[len(s) // 2 - 1] + s[len(s) // 2])


def percentile(v, q):
    """q in [0, 1]: nearest-rank percentile of a non-empty list."""
    s = sorted(v)
    return s[min(len(s) - 1, max(0, int(q * len(s))))]

import os
import sys

import pytest

# The C oracle (oracle/starkref.c) runs its loops under OpenMP on every host core.  On a GPU box whose cores are shared
# with other jobs, spinning idle OpenMP threads turn a 30 ms oracle call into 300 ms (seen: one test 3 s -> 385 s).  Idle
# threads sleep instead of spinning for the test run; bench.py's CPU-baseline legs are not affected (they do not come
# through here).  Must be set before libgomp starts, i.e. before the first oracle call of the process.
os.environ.setdefault("OMP_WAIT_POLICY", "PASSIVE")
os.environ.setdefault("GOMP_SPINCOUNT", "0")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "stark-perpetual_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")

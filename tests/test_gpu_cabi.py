"""The C ABI from plain C: compiles tests/cabi/cabi_smoke.c with gcc against
include/starkperp.h + libstarkperp.so and runs it on the GPU."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_c_consumer_runs():
    libdir = os.path.join(ROOT, "stark-perpetual_amd", "lib")
    exe = os.path.join(ROOT, "tests", "cabi", "cabi_smoke")
    subprocess.check_call(["gcc", "-O1", os.path.join(ROOT, "tests", "cabi", "cabi_smoke.c"), "-o", exe,
                           "-L" + libdir, "-lstarkperp", "-Wl,-rpath," + libdir])
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 0, (out.returncode, out.stdout, out.stderr)
    assert "cabi_smoke ok" in out.stdout

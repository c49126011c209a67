"""ISA-level check of the masked signer's claim (include/starkperp.h "Threading / threat model"): the walk
of csrc/masked_walk.hpp, compiled for gfx950 exactly as the library compiles it, has no control flow and no address
that depends on the secret scalar.  The probe kernel (tests/isa/masked_walk_probe.hip) contains nothing else that is
lane dependent, so the assertions can be absolute:

  * the only branch is the uniform loop back-edge (s_cbranch_scc*): nothing branches on VCC or EXEC;
  * EXEC is never written (no s_*_saveexec, no v_cmpx, no move into exec): no lane is ever masked off;
  * every table entry is fetched with SCALAR loads (s_load_*: one address for the whole wave, from the kernel's
    uniform table pointer and the loop counter) - the only vector loads are the two that fetch the lane's scalar;
  * no scratch / spill traffic (a spill address could be indexed).

This checks the walk in isolation; inside the signer kernels the same source is inlined behind an opaque-value barrier
on the mask (asm volatile), which is what keeps the compiler from re-deriving a branch there."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "stark-perpetual_amd", "csrc")


@pytest.fixture(scope="module")
def listing(tmp_path_factory):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc on this box")
    out = tmp_path_factory.mktemp("isa") / "masked_walk_probe.s"
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "--cuda-device-only", "-S", "-I" + CSRC,
           "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "isa", "masked_walk_probe.hip"), "-o", str(out)]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    text = out.read_text()
    a = text.index("masked_walk_probe:")
    b = text.index("s_endpgm", a)
    body = [l.split(";")[0].strip() for l in text[a:b].splitlines()]
    return [l for l in body if l and not l.startswith(".") and not l.endswith(":")]


def test_no_secret_dependent_control_flow(listing):
    branches = [l for l in listing if l.startswith(("s_cbranch", "s_branch", "s_setpc", "s_call"))]
    assert len(branches) == 1 and branches[0].startswith("s_cbranch_scc"), branches  # the 62-step loop, uniform
    exec_writes = [l for l in listing if "saveexec" in l or l.startswith("v_cmpx")
                   or re.match(r"s_\w+\s+exec(_lo|_hi)?\b", l)]
    assert exec_writes == [], exec_writes


def test_every_table_read_is_a_uniform_scalar_load(listing):
    vector_loads = [l for l in listing if re.match(r"(global|flat|buffer)_load", l)]
    assert len(vector_loads) == 2 and all(l.startswith("global_load_dwordx4") for l in vector_loads), vector_loads
    scalar_loads = [l for l in listing if l.startswith("s_load_dwordx16")]
    # 16 entries x 64 bytes per window = 16 s_load_dwordx16 per select(); select(0) + the loop body's select(i)
    assert len(scalar_loads) == 32, len(scalar_loads)
    assert [l for l in listing if l.startswith(("scratch_", "buffer_store", "ds_"))] == []


def test_the_signer_kernels_inline_this_header():
    src = open(os.path.join(CSRC, "ecdsa.hip")).read()
    assert '#include "masked_walk.hpp"' in src and "gen_mul_masked(" in src
    hdr = open(os.path.join(CSRC, "masked_walk.hpp")).read()
    assert 'asm volatile("" : "+v"(m))' in hdr

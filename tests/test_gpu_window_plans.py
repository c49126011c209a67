"""The benchmark's table configuration under the driver's pytest.  bench.py runs with 26-bit
windows (75 GiB of tables, 19 gathers per hash), the library default is 21 bits (4.3 GiB, 23 gathers): the full-size
parity cases - G1 (all 1024 reference hashes), C2 (the 2^16-leaf tree of the reference golden, every level), the
driver's own forest shape (20 x 2^16 leaves: all 20 roots and every node of tree 0 against the C oracle) and C3 (the
4096-order batch of g7) - run here under BOTH plans in one process (sp_shutdown + sp_init).  26 bits is skipped when
the device has less than 90 GiB free."""
import os

import pytest

import test_gpu_pedersen as TP
import test_gpu_state as TS
import workloads as wl

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", params=[21, 26], ids=["w21", "w26"])
def window_bits(request):
    import torch
    from starkperp import _lib
    wbits = request.param
    lib = _lib.load()
    if wbits == 26:
        free, _total = torch.cuda.mem_get_info(0)
        have = lib.sp_table_bytes() if lib.sp_is_initialised() else 0
        if free + have < 90 * 2**30:
            pytest.skip("26-bit tables need 75 GiB of HBM; %.0f GiB free" % ((free + have) / 2**30))
    torch.cuda.synchronize()
    lib.sp_shutdown()
    _lib.check(lib.sp_init(0, wbits), "sp_init")
    assert lib.sp_window_bits() == wbits
    yield wbits
    torch.cuda.synchronize()
    lib.sp_shutdown()
    _lib.ensure_init()  # back to the environment's plan for whatever runs next


@pytest.fixture(scope="module")
def batch():
    from starkperp import batch as b
    return b


def test_g1_full_batch(window_bits, batch):
    TP.test_g1_full_batch(batch)
    TP.test_edges_and_kats(batch)


def test_merkle_c2_full(window_bits, batch):
    TP.test_merkle_c2_full(batch)


def test_forest_of_the_driver_run(window_bits, batch):
    """20 lockstep 2^16-leaf trees through sp_merkle_forest_dev - the call bench.py times - against the C oracle:
    every root from the multi-threaded comparator, every inner node of tree 0 from the plain restatement."""
    import torch
    from oracle import cref
    from starkperp import _lib
    from starkperp.distributed import felt_to_tensor
    lib = _lib.ensure_init()
    nb, height = 20, 16
    n = 1 << height
    trees = [wl.leaves(n, seed=900 + t) for t in range(nb)]
    buf = torch.zeros((nb * (2 * n - 1), 4), dtype=torch.int64, device="cuda")
    flat = [v for t in trees for v in t]
    packed = _lib.pack_felts(flat)
    import numpy as np
    buf[: nb * n] = torch.from_numpy(np.frombuffer(packed, dtype="<i8").reshape(nb * n, 4).copy()).cuda()
    _lib.check(lib.sp_merkle_forest_dev(buf.data_ptr(), nb, height, None, torch.cuda.current_stream().cuda_stream), "forest")
    torch.cuda.synchronize()
    host = buf.cpu().numpy().astype("<i8")
    felts = _lib.unpack_felts((__import__("ctypes").c_uint64 * (4 * host.shape[0])).from_buffer_copy(host.tobytes()),
                              host.shape[0])
    # level j of the forest is one contiguous array: tree t owns [t * n >> j, (t + 1) * n >> j) of it
    offs, pos = [], 0
    for j in range(height + 1):
        offs.append(pos)
        pos += nb * (n >> j)
    want0 = cref.merkle_levels(trees[0])
    for j in range(height + 1):
        w = n >> j
        assert felts[offs[j]: offs[j] + w] == want0[j], "tree 0, level %d" % j
    for t in range(nb):
        assert felts[offs[height] + t] == cref.opt_merkle_levels(trees[t])[-1][0], "root of tree %d" % t


def test_c3_order_batch(window_bits):
    TS.test_c3_order_batch()

// ISA probe of the signer's masked walk (csrc/masked_walk.hpp): ONE kernel whose only secret-dependent input is the
// scalar it loads, no bounds check and no other lane-dependent control flow.  tests/test_masked_walk_isa.py compiles
// this file for gfx950 (device only, -S) and asserts on the listing: no instruction that writes EXEC from a lane
// value, no branch on VCC / EXEC, and every table load addressed from the kernel's uniform table pointer.
#include "masked_walk.hpp"

using namespace sp;

extern "C" __global__ void __launch_bounds__(128) masked_walk_probe(const uint64_t* __restrict__ k, uint64_t* __restrict__ out,
                                                                  const aff_packed* __restrict__ gen) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const u256 kk = ld_u256(k + 4 * i);
  const xyzz p = gen_mul_masked(kk, gen, 63);
  st_u256(out + 4 * i, fe_pack(fe_canon(p.X)));
  st_u256(out + 4 * (i + (size_t)gridDim.x * blockDim.x), fe_pack(fe_canon(p.ZZ)));
}

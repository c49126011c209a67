// The signer's masked fixed-base walk (STARKPERP_SIGN_MASKED=1), in a header of its own so that the ISA probe
// (tests/isa/masked_walk_probe.hip, compiled and disassembled by tests/test_masked_walk_isa.py) instantiates the
// SAME source the signer kernels inline.
#pragma once
#include "context.hpp"

namespace sp {

// k * EC_GEN with addresses and control flow that do not depend on k (STARKPERP_SIGN_MASKED=1; the signer's threat
// model in include/starkperp.h): `gen` is the table of 63 unsigned 4-bit windows (context.hpp gen_masked).  Window i
// reads ALL 16 of its entries - the same 16 addresses on every lane, whatever the scalar is - and keeps the one its
// nibble names with a mask built from a comparison of VALUES; the walk is the same 62 mixed additions for every k.
// About six times the work of the gathered walk below (12 additions at 21-bit windows), which is why it is opt-in.
__device__ __forceinline__ xyzz gen_mul_masked(u256 k, const aff_packed* __restrict__ gen, int nwin) {
  auto select = [&](int i) {
    const uint32_t v = k.w[0] & 15u;
#pragma unroll
    for (int w = 0; w < 7; ++w) k.w[w] = (k.w[w] >> 4) | (k.w[w + 1] << 28);
    k.w[7] >>= 4;
    uint32_t sel[16];
#pragma unroll
    for (int w = 0; w < 16; ++w) sel[w] = 0;
    const uint4* row = reinterpret_cast<const uint4*>(gen + (size_t)i * 16);
#pragma unroll
    for (uint32_t j = 0; j < 16; ++j) {
      uint32_t m = 0u - (uint32_t)(v == j);  // all ones for the wanted entry, zero otherwise
      // value barrier: from here on the compiler sees an opaque VGPR, not "the result of a comparison with the
      // secret nibble" - it cannot lower the masked ORs below into exec-masked / conditional loads or a branch on
      // the nibble.  tests/test_masked_walk_isa.py disassembles this walk and checks exactly that.
      asm volatile("" : "+v"(m));
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const uint4 t = row[4 * j + q];
        sel[4 * q + 0] |= t.x & m;
        sel[4 * q + 1] |= t.y & m;
        sel[4 * q + 2] |= t.z & m;
        sel[4 * q + 3] |= t.w & m;
      }
    }
    u256 x, y;
#pragma unroll
    for (int w = 0; w < 8; ++w) { x.w[w] = sel[w]; y.w[w] = sel[8 + w]; }
    aff a;
    a.x = fe_unpack(x);
    a.y = fe_unpack(y);
    return a;
  };
  xyzz acc = xyzz_from_aff(select(0));
#pragma unroll 1
  for (int i = 1; i < nwin; ++i) acc = xyzz_madd(acc, select(i));
  return acc;
}

}  // namespace sp

// Micro-benchmark: where the latency of one inversion goes on a lone wave (one wave per CU):
// fixed-length vs variable-time divsteps, the matrix application alone, a chain of multiplications.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../../stark-perpetual_amd/csrc lat_parts.hip -o lat_parts
#include <hip/hip_runtime.h>
#include <cstdio>
#include "fp29.hpp"
using namespace sp;

// MODE 7: the variable-time inversion of ONE value per wave with the divsteps batches on the SCALAR unit
// (f0, g0, eta and the transition matrix are wave-uniform: readfirstlane makes the compiler keep the whole
// batch loop in SGPRs / SALU instructions; only the matrix application stays on the VALU)
__device__ __forceinline__ fe inv_scalar_steps(const fe& x) {
  fe d = FE_ZERO, e = {{1, 0, 0, 0, 0, 0, 0, 0, 0}};
  fe f = FE_P, g = x;
  int32_t eta = -1;
  for (int it = 0; it < 26; ++it) {
    trans2x2 t;
    const uint32_t f0 = (uint32_t)__builtin_amdgcn_readfirstlane(f.l[0]);
    const uint32_t g0 = (uint32_t)__builtin_amdgcn_readfirstlane(g.l[0]);
    eta = divsteps_29_var(eta, f0, g0, t);
    gcd_update_de(d, e, t);
    gcd_update_fg(f, g, t);
    int32_t nz = 0;
#pragma unroll
    for (int i = 0; i < NL; ++i) nz |= g.l[i];
    if (__builtin_amdgcn_readfirstlane(nz) == 0) break;
  }
  const int32_t sf = f.l[NL - 1] >> 31;
  fe r;
#pragma unroll
  for (int i = 0; i < NL; ++i) r.l[i] = (d.l[i] ^ sf) - sf;
  return fe_carry(r);
}

// MODE 0: fe_inv_plain_gcd (fixed length)  1: fe_inv_plain_gcd_var  2: divsteps_29 x 18 only
//      3: divsteps_29_var x 18 only  4: gcd_update_de + gcd_update_fg x 18 only  5: 18 fe_mul  6: 18 fe_sqr
template <int MODE, bool SAME>
__global__ void __launch_bounds__(64) k(const int32_t* in, int32_t* out, int reps) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  fe a;
  for (int i = 0; i < NL; ++i) a.l[i] = (in[i] ^ (int32_t)(((SAME ? blockIdx.x : t) * 2654435761u) & 0xfffff)) & LMASK;
  a.l[8] &= 0x3ffff;
  fe acc = a;
  for (int r = 0; r < reps; ++r) {
    if (MODE == 0) acc = fe_inv_plain_gcd(fe_carry(fe_add(acc, a)));
    if (MODE == 1) acc = fe_inv_plain_gcd_var(fe_carry(fe_add(acc, a)));
    if (MODE == 2 || MODE == 3) {
      int32_t z = -1;
      for (int it = 0; it < 18; ++it) {
        trans2x2 tt;
        z = MODE == 2 ? divsteps_29(z, (uint32_t)acc.l[0] | 1u, (uint32_t)acc.l[1], tt)
                      : divsteps_29_var(z, (uint32_t)acc.l[0] | 1u, (uint32_t)acc.l[1], tt);
        acc.l[0] ^= tt.u & LMASK; acc.l[1] ^= tt.q & LMASK; acc.l[2] += tt.v & 0xff; acc.l[3] ^= tt.r & 0xff;
      }
    }
    if (MODE == 4) {
      fe d = acc, e = a, f = a, g = acc;
      trans2x2 tt = {acc.l[0] >> 1, acc.l[1] >> 2, -(acc.l[2] >> 1), acc.l[3] >> 2};
      for (int it = 0; it < 18; ++it) { gcd_update_de(d, e, tt); gcd_update_fg(f, g, tt); tt.u ^= d.l[0] >> 3; }
      acc = fe_carry(fe_add(fe_add(d, e), fe_add(f, g)));
      acc.l[8] &= 0x3ffff;
    }
    if (MODE == 7) acc = inv_scalar_steps(fe_carry(fe_add(acc, a)));
    if (MODE == 5) for (int it = 0; it < 18; ++it) acc = fe_mul(acc, a);
    if (MODE == 6) for (int it = 0; it < 18; ++it) acc = fe_sqr(acc);
  }
  for (int i = 0; i < NL; ++i) out[t * NL + i] = acc.l[i];
}

template <int MODE, bool SAME>
void run(const char* name, int blocks, int reps) {
  int32_t *in, *out;
  hipMalloc(&in, 64);
  hipMemset(in, 0x15, 64);
  hipMalloc(&out, (size_t)blocks * 64 * NL * 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  k<MODE, SAME><<<blocks, 64>>>(in, out, reps);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  k<MODE, SAME><<<blocks, 64>>>(in, out, reps);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  printf("%-64s %5d waves: %7.2f us per repetition\n", name, blocks, ms * 1e3 / reps);
  hipFree(in); hipFree(out);
}

int main() {
  const int reps = 8;
  for (int blocks : {256, 2048}) {
    run<0, false>("fixed-length inversion, distinct values per lane", blocks, reps);
    run<1, false>("variable-time inversion, distinct values per lane", blocks, reps);
    run<1, true>("variable-time inversion, one value per wave", blocks, reps);
    run<7, true>("variable-time inversion, one value per wave, divsteps on the SALU", blocks, reps);
    run<2, false>("18 x divsteps_29 (fixed)", blocks, reps);
    run<3, false>("18 x divsteps_29_var, distinct", blocks, reps);
    run<3, true>("18 x divsteps_29_var, one value per wave", blocks, reps);
    run<4, false>("18 x (gcd_update_de + gcd_update_fg)", blocks, reps);
    run<5, false>("18 x fe_mul", blocks, reps);
    run<6, false>("18 x fe_sqr", blocks, reps);
  }
  return 0;
}

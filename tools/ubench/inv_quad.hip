// Micro-benchmark + check: the quad-split inversion of the latency path, divsteps form (round 2) against the
// double-steered Lehmer form (round 3), on lone waves (one wave per CU) and with the chip's SIMDs all busy.
// Every result is checked: both forms must return the same canonical inverse, and x * x^-1 = 1.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../../stark-perpetual_amd/csrc inv_quad.hip -o inv_quad
#include <hip/hip_runtime.h>
#include <cstdio>
#include "quad.hpp"
using namespace sp;

// MODE 0: divsteps quad  1: lehmer quad  2: lehmer lane-private  3: divsteps lane-private (variable time)
template <int MODE, bool SAME>
__global__ void __launch_bounds__(64) k(const int32_t* in, int32_t* out, int reps) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t grp = MODE < 2 ? (SAME ? blockIdx.x * 16 : t >> 2) : (SAME ? blockIdx.x : t);
  fe a;
  for (int i = 0; i < NL; ++i) a.l[i] = (in[i] ^ (int32_t)(((grp + 1) * 2654435761u * (i + 1)) & 0xfffffff)) & LMASK;
  a.l[8] &= 0x3ffff;
  fe acc = a;
  const int kq = (int)(threadIdx.x & 3);
  for (int r = 0; r < reps; ++r) {
    fe x = fe_carry(fe_add(acc, a));
    x.l[8] &= 0x3ffff;
    if (MODE == 0) acc = fe_inv_plain_quad_divsteps(x, kq);
    if (MODE == 1) acc = fe_inv_plain_quad(x, kq);
    if (MODE == 2) acc = fe_inv_plain_lehmer(x);
    if (MODE == 3) acc = fe_inv_plain_gcd_var(x);
  }
  for (int i = 0; i < NL; ++i) out[t * NL + i] = acc.l[i];
}

// correctness: inverse by both quad forms and the lane-private ones, product with the input must be one
__global__ void __launch_bounds__(64) check(const int32_t* in, int* bad, int special) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t grp = t >> 2;
  fe a;
  for (int i = 0; i < NL; ++i) a.l[i] = (in[i] ^ (int32_t)(((grp + 7) * 2654435761u * (i + 3)) & 0xfffffff)) & LMASK;
  a.l[8] &= 0x3ffff;
  if (special) {  // small values, powers of two, p - small: the fallback path
    const int sel = (int)(grp % 6);
    fe s = FE_ZERO;
    if (sel == 0) s.l[0] = (int32_t)(grp & 0xffff) + 1;
    if (sel == 1) s.l[(grp >> 3) % 9] = 1 << ((grp >> 7) % 19);
    if (sel == 2) { s = FE_P; s.l[0] -= 0; s.l[1] = 0; s = fe_carry(fe_sub(s, a)); s.l[8] &= 0x3ffff; }
    if (sel == 3) { s = a; for (int i = 3; i < NL; ++i) s.l[i] = 0; }
    if (sel == 4) { s = FE_P; s.l[0] = 0; }  // p - 1
    if (sel == 5) s = a;
    a = s;
  }
  const int kq = (int)(threadIdx.x & 3);
  const fe canon = fe_canon(a);
  const fe i0 = fe_inv_plain_quad_divsteps(canon, kq), i1 = fe_inv_plain_quad(canon, kq);
  const fe i2 = fe_inv_plain_lehmer(canon), i3 = fe_inv_plain_gcd_var(canon);
  // unreduced representatives (x + 3p, x - 2p: what the callers hand over in N-form) must give the same inverse
  const fe i4 = fe_inv_plain_quad(fe_carry(fe_add(fe_add(canon, FE_P), fe_dbl(FE_P))), kq);
  const fe i5 = fe_inv_plain_quad(fe_carry(fe_sub(canon, fe_dbl(FE_P))), kq);
  int diff = 0;
  for (int i = 0; i < NL; ++i)
    diff |= (i0.l[i] ^ i1.l[i]) | (i0.l[i] ^ i2.l[i]) | (i0.l[i] ^ i3.l[i]) | (i0.l[i] ^ i4.l[i]) | (i0.l[i] ^ i5.l[i]);
  // x * x^-1 == 1 (plain values: to Montgomery, multiply, back)
  const fe prod = fe_from_mont(fe_mul(fe_to_mont(canon), fe_to_mont(i1)));
  int one = prod.l[0] ^ 1;
  for (int i = 1; i < NL; ++i) one |= prod.l[i];
  int zero = 0;
  for (int i = 0; i < NL; ++i) zero |= canon.l[i];
  if (diff != 0 || (zero != 0 && one != 0)) atomicAdd(bad, 1);
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
  printf("%-72s %5d waves: %7.2f us per inversion\n", name, blocks, ms * 1e3 / reps);
  hipFree(in); hipFree(out);
}

int main() {
  int32_t* in; int* bad;
  hipMalloc(&in, 64); hipMemset(in, 0x15, 64);
  hipMalloc(&bad, 4); hipMemset(bad, 0, 4);
  check<<<4096, 64>>>(in, bad, 0);
  check<<<1024, 64>>>(in, bad, 1);
  int h = -1;
  hipMemcpy(&h, bad, 4, hipMemcpyDeviceToHost);
  printf("check: %d mismatching lanes over 262144 random + 65536 special lanes (four inversion forms + unreduced inputs, x * x^-1 == 1)\n", h);
  const int reps = 8;
  for (int blocks : {256, 1024, 2048}) {
    run<0, true>("quad-split divsteps, one value per wave", blocks, reps);
    run<1, true>("quad-split lehmer (f64), one value per wave", blocks, reps);
    run<0, false>("quad-split divsteps, 16 values per wave", blocks, reps);
    run<1, false>("quad-split lehmer (f64), 16 values per wave", blocks, reps);
    run<3, false>("lane-private divsteps (variable time), 64 values per wave", blocks, reps);
    run<2, false>("lane-private lehmer (f64), 64 values per wave", blocks, reps);
  }
  return h == 0 ? 0 : 1;
}
